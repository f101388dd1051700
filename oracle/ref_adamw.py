"""TEST INFRASTRUCTURE ONLY (see oracle/vd_oracle.py header): restatement of the optimizer train_gen.py:16,247 imports,
`pytorch_transformers==1.2.0` `optimization.AdamW` (third-party, pinned in the reference's requirements, not vendored in
/root/reference and not installed here).  Its published algorithm, per parameter with a gradient:

    state: step (int), exp_avg, exp_avg_sq              (created lazily at the first step that sees a gradient)
    exp_avg    = b1 * exp_avg    + (1 - b1) * g
    exp_avg_sq = b2 * exp_avg_sq + (1 - b2) * g * g
    step_size  = lr * sqrt(1 - b2^step) / (1 - b1^step)          (correct_bias=True)
    p         -= step_size * exp_avg / (sqrt(exp_avg_sq) + eps)   (eps OUTSIDE the bias correction, default 1e-6)
    p         -= lr * weight_decay * p                            (decoupled, AFTER the Adam update, using the group's lr)

It subclasses torch.optim.Optimizer exactly like the original, so `state_dict()` has the on-disk layout the reference's
checkpoints carry in 'optimizer_state_dict' (train_gen.py:349): {'state': {index: {...}}, 'param_groups': [...]} with one
param group per tensor (train_gen.py:209-245).  Used by oracle/make_golden_r2.py on the REAL reference model and by the tests
as the checker of gst_visdial_amd.optim / checkpoint."""
import math

import torch
from torch.optim import Optimizer


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad.data
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p.data)
                    state["exp_avg_sq"] = torch.zeros_like(p.data)
                exp_avg, exp_avg_sq = state["exp_avg"], state["exp_avg_sq"]
                b1, b2 = group["betas"]
                state["step"] += 1
                exp_avg.mul_(b1).add_(grad, alpha=1.0 - b1)
                exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
                denom = exp_avg_sq.sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - b2 ** state["step"]) / (1.0 - b1 ** state["step"])
                p.data.addcdiv_(exp_avg, denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.data.add_(p.data, alpha=-group["lr"] * group["weight_decay"])
        return None


NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def reference_param_groups(dialog_encoder, dialog_decoder, lr, image_lr, language_weights):
    """train_gen.py:204-245: one group per tensor, encoder tensors first, then the decoder's (built BEFORE the embedding
    aliasing of train_gen.py:293, so the decoder's own -- later orphaned -- embedding tensors are in the list).
    Returns (groups, names) with names relative to each module, as the reference's loops see them."""
    groups, names = [], []
    for prefix, mod in (("encoder", dialog_encoder), ("decoder", dialog_decoder)):
        for key, value in dict(mod.named_parameters()).items():
            if not value.requires_grad:
                continue
            g_lr = lr if (language_weights is None or key in language_weights) else image_lr
            wd = 0.0 if any(nd in key for nd in NO_DECAY) else 0.01
            groups.append({"params": [value], "lr": g_lr, "weight_decay": wd})
            names.append(prefix + "." + key)
    return groups, names
