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


def reference_param_index(model):
    """[(name, parameter | None)] in the order the reference builds its per-tensor param groups (train_gen.py:209-245):
    `dialog_encoder.named_parameters()` then `dialog_decoder.named_parameters()`, both taken BEFORE the embedding aliasing
    of train_gen.py:293.  Two consequences of that order, both reproduced here so that group/state indices line up with
    the reference's `optimizer.state_dict()`:
      * the decoder's own embedding tensors are in the list; after the aliasing they are orphans that never see a
        gradient (-> None here: no state), EXCEPT its word embedding, which the LM head keeps as its weight
        (visual_dialog_decoder.py:124,329-335): that slot is `decoder.lm_head.decoder.weight`;
      * `lm_head.decoder.weight` itself is not listed a second time (named_parameters de-duplicates the tied tensor)."""
    dec_mod = model.decoder.decoder
    if dec_mod.bert.embeddings is not model.encoder.bert_pretrained.bert.embeddings:
        raise NotImplementedError("reference optimizer state is defined for the train_gen.py set-up: share the embeddings "
                                  "first (decoder.decoder.bert.embeddings = encoder.bert_pretrained.bert.embeddings)")
    out = [("encoder." + k, p) for k, p in model.encoder.named_parameters()]
    lm_w = dec_mod.lm_head.decoder.weight
    for k, p in model.decoder.named_parameters():
        if p is lm_w:
            continue
        if k.startswith("decoder.bert.embeddings."):
            p = lm_w if k == "decoder.bert.embeddings.word_embeddings.weight" else None
        out.append(("decoder." + k, p))
    return out


class _FuseHandle(object):
    """See FusedAdamW.fuse_handle()."""

    def __init__(self, opt, write_grad):
        from . import _lib as L
        self.opt, self.write_grad = opt, write_grad
        self._desc = L.AdamFuse()
        self._g0 = opt.engine.flat.G.data_ptr()

    def desc(self):
        """The launch's view of the optimizer, refreshed from the live tensors / constants on every call."""
        opt, d = self.opt, self._desc
        flat = opt.engine.flat
        d.grad_base, d.param, d.m, d.v = flat.G.data_ptr(), flat.P.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr()
        d.shadow_bf16 = flat.S.data_ptr() if flat.S is not None else None
        d.step, d.beta1, d.beta2, d.eps = opt.step_dev.data_ptr(), opt.betas[0], opt.betas[1], opt.eps
        d.grad_scale, d.write_grad = opt.grad_scale, int(self.write_grad)
        if d.grad_base != self._g0:
            raise RuntimeError("FusedAdamW: the flat gradient buffer moved under a live fuse handle")
        return d

    def cover(self, c_ptr, M, N, ldc):
        """(device address of the (lr, wd) pair, flat offsets of the tensors) of the weight(s) whose gradient slot the contiguous
        [M, N] block at c_ptr is, or (0, ()) when the launch must not update it.  The block may span several consecutive tensors
        with the same (lr, wd) -- the fused Q|K|V projections are ONE weight-gradient problem over three tensors -- and may end
        in alignment padding (the LM head's rows are padded to a multiple of 64: gradient, moments and weights of the padding
        are zero and stay zero under any lr).  Frozen tensors and blocks that stop inside a tensor are left to the plain path."""
        opt = self.opt
        off = c_ptr - self._g0
        if off < 0 or off % 16 or ldc != N or N % 4:
            return 0, ()
        off //= 4
        first = opt.seg_of.get(off)
        if first is None:
            return 0, ()
        end, i, offs, hp0 = off + M * N, first[0], [], opt.base[first[0]]
        if not hp0[2]:
            return 0, ()
        pos = off
        while pos < end and i < len(opt.seg_ends_host):
            live = opt.base[i][2]
            if live:
                if opt.base[i] != hp0 or pos not in opt.seg_of:
                    return 0, ()
                offs.append(pos)
            elif pos in opt.seg_of or opt.seg_ends_host[i] < end:
                return 0, ()                                # a frozen tensor, or padding in the MIDDLE of the block
            pos = opt.seg_ends_host[i]
            i += 1
        if pos < end or (pos > end and opt.base[i - 1][2]):
            return 0, ()                                    # stops inside a live tensor
        return opt.hp.data_ptr() + 8 * first[0], tuple(offs)


class FusedAdamW(object):
    def __init__(self, model, lr=2e-5, image_lr=None, language_weights=None, weight_decay=0.01, betas=(0.9, 0.999),
                 eps=1e-6, warmup_steps=0, t_total=0, min_lr=1e-5, train_vlfusion=False):
        self.model, self.engine = model, model.engine
        self.betas, self.eps = betas, eps
        self.lr = lr
        self.image_lr = lr if image_lr is None else image_lr
        self.language_weights = set(language_weights) if language_weights is not None else None
        self.weight_decay = weight_decay
        self.warmup_steps, self.t_total, self.min_lr = warmup_steps, t_total, min_lr
        # Reference quirk, preserved by default: train_gen.py:209-245 builds its param groups from dialog_encoder and
        # dialog_decoder only -- EncoderDecoderModel.vlfusion (visual_dialog_model.py:22) is in neither, so the reference never
        # updates fc_v / fc_l (golden: tests/golden/tiny_trainer.npz keeps them bit-identical over 6 iterations).
        self.train_vlfusion = train_vlfusion
        self.sched_step = 0          # scheduler.step() count (train_gen.py:329 steps it every iteration)
        self.opt_step = 0            # optimizer.step() count
        self.grad_scale = 1.0        # e.g. 1/world_size after a summing all-reduce
        self._built = False
        self._applied_in_backward = False
        self._pending = None         # a state dict loaded before the flat buffers existed (train_gen.py:254-276 loads first)

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
        ends, base, seg_of = [], [], {}
        prev = 0
        for p, off in flat.items:
            if off > prev:                       # alignment gap / zero padding: lr 0 keeps it untouched
                ends.append(off); base.append((0.0, 0.0, 0.0))
            n = names[id(p)]
            lr = self.lr if (self.language_weights is None or n in self.language_weights) else self.image_lr
            wd = 0.0 if any(nd in n for nd in NO_DECAY) else self.weight_decay
            frozen = n.startswith("vlfusion.") and not self.train_vlfusion
            seg_of[off] = (len(ends), p.numel())
            ends.append(off + p.numel()); base.append((0.0, 0.0, 0.0) if frozen else (lr, wd, 1.0))
            prev = off + p.numel()
        if prev < flat.n_live:
            ends.append(flat.n_live); base.append((0.0, 0.0, 0.0))
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
        self.seg_ends_host, self.seg_of = ends, seg_of      # flat offset of a tensor -> (segment index, numel)
        self._fuse, self._remainders = None, {}
        self.base = base
        self.hp_host = torch.empty(len(base) * 2, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.empty(len(base) * 2)
        self.hp = torch.empty(len(base) * 2, dtype=torch.float32, device=dev)
        if self._pending is None:              # (a rebuild after replayed steps: the device counter is the truth -- unless a
            self._sync_step()                  #  checkpoint is queued: its opt_step must not be overwritten by the stale counter)
        old = (self.m, self.v) if self._built and self.m.numel() == flat.P.numel() else None
        self.m = torch.zeros_like(flat.P)
        self.v = torch.zeros_like(flat.P)
        if old is not None:                      # the flat buffers were re-materialised (model.to(), aliasing): keep the moments
            self.m.copy_(old[0]); self.v.copy_(old[1])
        # bias correction reads this counter on the device (loss.hip adamw_kernel): it must follow opt_step across resumes
        self.step_dev = torch.full((1,), float(self.opt_step), dtype=torch.float32, device=dev)
        self._built, self._flat_id = True, id(flat.P)
        self._last_lr_key = None
        if self._pending is not None:
            sd, self._pending = self._pending, None
            self._apply_state(sd)

    def _sync_step(self):
        """optimizer.step() count: the DEVICE counter is the source of truth.  Under hipGraph replay `begin_step()` runs once,
        at capture; the captured `step_dev += 1` then advances on the device with every replay while the host copy stands
        still -- a checkpoint written from the host copy would restart AdamW's bias correction near t = 1 on warm moments."""
        if self._built:
            if self.step_dev.is_cuda and torch.cuda.is_current_stream_capturing():
                # .item() is a device synchronisation: inside a stream capture it would invalidate the capture (and, at N>1, take
                # the c10d watchdog down with it).  A rebuild of the optimizer state belongs in front of the capture.
                raise RuntimeError("FusedAdamW: the optimizer state has to be (re)built while a hipGraph capture is running (the flat "
                                   "parameter buffer was re-materialised?) -- run one eager step, or call state_dict(), before capturing")
            self.opt_step = int(round(float(self.step_dev.item())))
        return self.opt_step

    def _ensure_built(self):
        flat = self.engine.flat
        if flat is None or flat.P is None:
            return False
        if not self._built or self._flat_id != id(flat.P):
            self._build()
        return True

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
        if not self._ensure_built():
            raise RuntimeError("FusedAdamW before the first forward")
        self._upload_hp()                 # host->device only when the learning rate changed (never inside a captured graph)
        self.opt_step += 1
        self.step_dev.add_(1.0)

    def apply_range(self, lo, hi, grad_bf16=None, fused=(), grad_origin=None):
        """AdamW on flat elements [lo, hi) -- used slice by slice by the backward pipeline.  `grad_bf16`: the slice's
        gradients as a bf16 tensor of hi-lo elements (the all-reduced compressed copy) instead of G[lo:hi].  `fused`: flat
        offsets of the weights the slice's weight-gradient launch has already updated (ops.GemmGroup.flush(fuse=...)): only the
        1024-element blocks that hold anything else are visited, and those weights' segments are skipped inside them."""
        flat = self.engine.flat
        if fused:
            if grad_bf16 is not None:
                raise ValueError("a fused weight-gradient update and a compressed gradient slice exclude each other")
            blocks, skip = self._remainder(lo, hi, fused)
            if blocks.numel():
                ops.adamw_blocks(flat.P, flat.G, self.m, self.v, flat.S, self.seg_end, self.hp, self.step_dev, blocks, skip,
                                 self.betas[0], self.betas[1], self.eps, self.grad_scale, begin=lo, end=hi)
        elif grad_bf16 is not None:
            # grad_bf16[k] is the gradient of flat element grad_origin + k (default: the tensor starts at lo)
            ops.adamw(flat.P, grad_bf16, self.m, self.v, flat.S, self.seg_end, self.hp, self.step_dev, self.betas[0],
                      self.betas[1], self.eps, self.grad_scale, begin=lo, end=hi, grad_origin=lo if grad_origin is None else grad_origin)
        else:
            ops.adamw(flat.P, flat.G, self.m, self.v, flat.S, self.seg_end, self.hp, self.step_dev, self.betas[0],
                      self.betas[1], self.eps, self.grad_scale, begin=lo, end=hi)
        if flat.S is not None:
            flat.shadow_version = flat.version()

    def _remainder(self, lo, hi, fused):
        """(block list, segment skip flags) of apply_range(lo, hi, fused=...), cached per (lo, hi, fused) -- built once, before any
        hipGraph capture (the eager warm-up steps run the same slices)."""
        key = (lo, hi, fused)
        hit = self._remainders.get(key)
        if hit is None:
            if self.seg_end.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdamW: new slice / fusion pattern during a hipGraph capture -- run one eager step first")
            dev = self.seg_end.device
            fused_segs = set(self.seg_of[o][0] for o in fused)
            skip = torch.zeros(len(self.seg_ends_host), dtype=torch.uint8)
            need = set()
            start = 0
            for i, end in enumerate(self.seg_ends_host):
                a, b = max(start, lo), min(end, hi)
                start = end
                if i in fused_segs:
                    skip[i] = 1
                    continue
                if b <= a or not self.base[i][2]:          # outside the slice / padding or frozen (lr 0)
                    continue
                need.update(range(a // 1024, (b - 1) // 1024 + 1))
            blocks = torch.tensor(sorted(need), dtype=torch.int32)
            hit = (blocks.to(dev), skip.to(dev))
            if len(self._remainders) > 64:
                self._remainders.clear()
            self._remainders[key] = hit
        return hit

    def fuse_handle(self, write_grad=False):
        """What ops.GemmGroup.flush(fuse=...) needs to let the weight-gradient launch run this optimizer's update in its epilogue
        (single GPU, gradients not accumulated over several backward passes): the flat buffers + constants, and the device address
        of a weight's (lr, wd) pair.  None until the optimizer state exists."""
        if not self._ensure_built():
            return None
        if self._fuse is None or self._fuse.write_grad != bool(write_grad):
            self._fuse = _FuseHandle(self, bool(write_grad))
        return self._fuse

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
        if self._pending is not None:          # loaded before the first forward and not applied yet: hand it back unchanged
            if "per_tensor" in self._pending:
                return dict(reference_state=self._pending["osd"], sched_step=self.sched_step)
            return dict(self._pending)
        if not self._built:
            return {}
        pipe = getattr(self.engine, "pipe", None)
        if pipe is not None and hasattr(pipe, "check_master_current"):
            pipe.check_master_current("optimizer.state_dict()")
        return dict(m=self.m, v=self.v, opt_step=self._sync_step(), sched_step=self.sched_step)

    def load_state_dict(self, sd):
        """Usable BEFORE the first forward, as train_gen.py:254-276 does it (checkpoint first, then training): the state is
        kept and applied as soon as the flat buffers exist.  Restores the device step counter too -- AdamW's bias correction
        sqrt(1-b2^t)/(1-b1^t) must continue at t = opt_step, not restart at 1 on warm moments."""
        if "reference_state" in sd:   # what state_dict() returns for a reference-format state that was never applied
            self.sched_step = int(sd.get("sched_step", self.sched_step))
            return self.import_reference_state(sd["reference_state"])
        if not sd:
            return
        self.opt_step, self.sched_step = int(sd["opt_step"]), int(sd["sched_step"])
        self._pending = sd
        self._ensure_built()         # applies it now when the engine is ready; otherwise begin_step() will
        if self._built and self._pending is not None:
            sd, self._pending = self._pending, None
            self._apply_state(sd)

    # ---- the reference's on-disk optimizer layout (train_gen.py:349 saves optimizer.state_dict(), :254-276 loads it) ----
    def export_reference_state(self):
        """-> {'state': {index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}: what the reference's
        pytorch_transformers AdamW (a torch.optim.Optimizer) would hold after the same steps -- one group per tensor in
        `reference_param_index` order, state only for tensors that have received gradients."""
        if not self._ensure_built():
            raise RuntimeError("FusedAdamW before the first forward")
        pipe = getattr(self.engine, "pipe", None)
        if pipe is not None and hasattr(pipe, "check_master_current"):
            pipe.check_master_current("optimizer.export_reference_state()")
        flat = self.engine.flat
        off_of = {id(p): off for p, off in flat.items}
        names = self._names()
        lr_t, lr_i = self.current_lrs()
        self._sync_step()
        state, groups = {}, []
        for i, (name, p) in enumerate(reference_param_index(self.model)):
            key = names.get(id(p), name) if p is not None else name.split(".", 1)[1]
            base = self.lr if (self.language_weights is None or key in self.language_weights) else self.image_lr
            wd = 0.0 if any(nd in key for nd in NO_DECAY) else self.weight_decay
            groups.append(dict(lr=(lr_t if base == self.lr else lr_i), initial_lr=base, betas=tuple(self.betas), eps=self.eps,
                               weight_decay=wd, correct_bias=True, params=[i]))
            if p is not None and id(p) in off_of and self.opt_step > 0:
                o, n = off_of[id(p)], p.numel()
                state[i] = dict(step=self.opt_step, exp_avg=self.m[o:o + n].view(p.shape).clone(),
                                exp_avg_sq=self.v[o:o + n].view(p.shape).clone())
        return dict(state=state, param_groups=groups)

    def import_reference_state(self, osd):
        """Load a reference-format optimizer state (see export_reference_state); usable before the first forward."""
        idx = reference_param_index(self.model)
        st = osd["state"]
        step = max([int(v["step"]) for v in st.values()] or [0])
        pend = []
        for i, (name, p) in enumerate(idx):
            e = st.get(i, st.get(str(i)))
            if e is None or p is None:
                continue
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise RuntimeError("optimizer state %d (%s): shape %s vs parameter %s" % (i, name, tuple(e["exp_avg"].shape), tuple(p.shape)))
            pend.append((p, e["exp_avg"], e["exp_avg_sq"]))
        self.opt_step = step
        self._pending = dict(per_tensor=pend, opt_step=step, sched_step=self.sched_step, osd=osd)
        if self._ensure_built() and self._pending is not None:
            sd, self._pending = self._pending, None
            self._apply_state(sd)

    def _apply_state(self, sd):
        if "per_tensor" in sd:
            off_of = {id(p): off for p, off in self.engine.flat.items}
            self.m.zero_(); self.v.zero_()
            for p, ea, eas in sd["per_tensor"]:
                o, n = off_of[id(p)], p.numel()
                self.m[o:o + n].copy_(ea.reshape(-1)); self.v[o:o + n].copy_(eas.reshape(-1))
            self.opt_step = int(sd["opt_step"])
            self.step_dev.fill_(float(self.opt_step))
            self._last_lr_key = None
            return
        if sd["m"].numel() != self.m.numel():
            raise RuntimeError("optimizer state holds %d elements, the model's flat buffer %d" % (sd["m"].numel(), self.m.numel()))
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.opt_step, self.sched_step = int(sd["opt_step"]), int(sd["sched_step"])
        self.step_dev.fill_(float(self.opt_step))
        self._last_lr_key = None
