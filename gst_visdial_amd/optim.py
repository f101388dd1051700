"""Fused AdamW + LR schedule for the flat parameter buffer.

Semantics follow what train_gen.py:204-247,326-329 builds: `pytorch_transformers==1.2.0` AdamW (eps added to
sqrt(v), bias correction, decoupled weight decay applied after the Adam update), one (lr, weight_decay) pair
per tensor (decay 0 for bias / LayerNorm tensors, else 0.01; `lr` vs `image_lr` selected per tensor name), and
`WarmupLinearScheduleNonZero` (utils/optim_utils.py:8-26).  One kernel launch updates all ~372 M live
parameters and refreshes the bf16 shadow weights in the same pass.
"""
import torch

from . import ops

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def warmup_linear_nonzero(step, warmup_steps, t_total, base_lr, min_lr=1e-5):
    """utils/optim_utils.py:19-26."""
    if step < warmup_steps:
        f = float(step) / float(max(1, warmup_steps))
    else:
        f = max(0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))
    return base_lr * f if base_lr * f > min_lr else min_lr


class FusedAdamW(object):
    def __init__(self, model, lr=2e-5, image_lr=None, language_weights=None, weight_decay=0.01, betas=(0.9, 0.999),
                 eps=1e-6, warmup_steps=0, t_total=0, min_lr=1e-5):
        self.model, self.engine = model, model.engine
        self.betas, self.eps = betas, eps
        self.lr = lr
        self.image_lr = lr if image_lr is None else image_lr
        self.language_weights = set(language_weights) if language_weights is not None else None
        self.weight_decay = weight_decay
        self.warmup_steps, self.t_total, self.min_lr = warmup_steps, t_total, min_lr
        self.sched_step = 0          # scheduler.step() count (train_gen.py:329 steps it every iteration)
        self.opt_step = 0            # optimizer.step() count
        self.grad_scale = 1.0        # e.g. 1/world_size after a summing all-reduce
        self._built = False
        self._applied_in_backward = False

    def _names(self):
        """Parameter names relative to the encoder / decoder module, as train_gen.py:211,229 sees them."""
        names = {}
        for n, p in self.model.encoder.named_parameters():
            names.setdefault(id(p), n)
        for n, p in self.model.decoder.named_parameters():
            names.setdefault(id(p), n)
        for n, p in self.model.named_parameters():
            names.setdefault(id(p), n)
        return names

    def _build(self):
        flat = self.engine.flat
        dev = flat.P.device
        names = self._names()
        ends, base = [], []
        prev = 0
        for p, off in flat.items:
            if off > prev:                       # alignment gap / zero padding: lr 0 keeps it untouched
                ends.append(off); base.append((0.0, 0.0, 0.0))
            n = names[id(p)]
            lr = self.lr if (self.language_weights is None or n in self.language_weights) else self.image_lr
            wd = 0.0 if any(nd in n for nd in NO_DECAY) else self.weight_decay
            ends.append(off + p.numel()); base.append((lr, wd, 1.0))
            prev = off + p.numel()
        if prev < flat.n_live:
            ends.append(flat.n_live); base.append((0.0, 0.0, 0.0))
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
        self.base = base
        self.hp_host = torch.empty(len(base) * 2, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.empty(len(base) * 2)
        self.hp = torch.empty(len(base) * 2, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(flat.P)
        self.v = torch.zeros_like(flat.P)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._built, self._flat_id = True, id(flat.P)
        self._last_lr_key = None

    def current_lrs(self):
        if self.t_total > 0:
            return (warmup_linear_nonzero(self.sched_step, self.warmup_steps, self.t_total, self.lr, self.min_lr),
                    warmup_linear_nonzero(self.sched_step, self.warmup_steps, self.t_total, self.image_lr, self.min_lr))
        return self.lr, self.image_lr

    def _upload_hp(self):
        lr_t, lr_i = self.current_lrs()
        key = (lr_t, lr_i)
        if key == self._last_lr_key:
            return
        for i, (lr, wd, live) in enumerate(self.base):
            cur = 0.0 if not live else (lr_t if lr == self.lr else lr_i)
            self.hp_host[2 * i] = cur
            self.hp_host[2 * i + 1] = wd
        self.hp.copy_(self.hp_host, non_blocking=True)
        self._last_lr_key = key

    def begin_step(self):
        """Advance the step counter (device resident: correct under hipGraph replay) and make sure the lr table is current."""
        flat = self.engine.flat
        if flat is None or flat.P is None:
            raise RuntimeError("FusedAdamW before the first forward")
        if not self._built or self._flat_id != id(flat.P):
            self._build()
        self._upload_hp()                 # host->device only when the learning rate changed (never inside a captured graph)
        self.opt_step += 1
        self.step_dev.add_(1.0)

    def apply_range(self, lo, hi, grad_bf16=None):
        """AdamW on flat elements [lo, hi) -- used slice by slice by the backward pipeline.  `grad_bf16`: the slice's
        gradients as a bf16 tensor of hi-lo elements (the all-reduced compressed copy) instead of G[lo:hi]."""
        flat = self.engine.flat
        if grad_bf16 is not None:
            ops.adamw(flat.P, grad_bf16, self.m, self.v, flat.S, self.seg_end, self.hp, self.step_dev, self.betas[0],
                      self.betas[1], self.eps, self.grad_scale, begin=lo, end=hi, grad_origin=lo)
        else:
            ops.adamw(flat.P, flat.G, self.m, self.v, flat.S, self.seg_end, self.hp, self.step_dev, self.betas[0],
                      self.betas[1], self.eps, self.grad_scale, begin=lo, end=hi)
        if flat.S is not None:
            flat.shadow_version = flat.version()

    def step(self):
        """optimizer.step(): one fused launch over the flat buffers -- or nothing, when a BackwardPipeline already
        applied this step's update slice by slice during loss.backward()."""
        if self._applied_in_backward:
            self._applied_in_backward = False
            return
        self.begin_step()
        self.apply_range(0, self.engine.flat.n_live)

    def upload_lr(self):
        """Refresh the device learning-rate table (call between hipGraph replays when the schedule moved)."""
        if self._built:
            self._upload_hp()

    def scheduler_step(self):
        self.sched_step += 1

    def zero_grad(self):
        for p in self.engine.flat.live:
            p.grad = None

    def state_dict(self):
        return dict(m=self.m, v=self.v, opt_step=self.opt_step, sched_step=self.sched_step) if self._built else {}

    def load_state_dict(self, sd):
        if not self._built:
            self._build()
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.opt_step, self.sched_step = int(sd["opt_step"]), int(sd["sched_step"])
