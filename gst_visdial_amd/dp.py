"""Data-parallel gradient synchronisation: one process per GPU, RCCL all-reduce over xGMI.

The reference trains with single-process `nn.DataParallel` (train_gen.py:295): per step it broadcasts all
388 M parameters, gathers the [B,25,30522] logits and reduce-adds gradients onto GPU 0.  Here every rank keeps
its parameters resident, and the only exchange is a sum all-reduce of the flat gradient buffer, issued in a
few large contiguous slices while backward is still running: the engine calls `hook(off)` when every gradient
at flat offset >= off is final (backward finishes the flat buffer from the end to the start), and each call
launches an asynchronous all-reduce of the newly completed region once it exceeds `bucket_elems`.
The 1/world_size scale is folded into the optimizer (`FusedAdamW.grad_scale`), so loss = mean over ranks of
the per-rank token mean, exactly DataParallel's gather + `.mean()` (train_gen.py:134-135).

xGMI is point to point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so buckets are large
(default 64 Mi elements = 256 MB fp32) -- 6 collectives per step instead of hundreds.
"""
import torch
import torch.distributed as dist


class GradSync(object):
    def __init__(self, engine, group=None, bucket_elems=64 << 20, compress=None):
        self.engine, self.group = engine, group
        self.bucket_elems = bucket_elems
        self.compress = compress          # None | "bf16"
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.handles = []
        self.hi = None
        self.slices = []                  # (lo, hi) log of the last step, for tests / DESIGN.md
        if self.world > 1:
            engine.grad_hook = self.hook

    def begin(self):
        self.hi = self.engine.flat.n_live
        self.handles, self.slices = [], []

    def hook(self, off):
        if self.world == 1:
            return
        if self.hi is None:
            self.begin()
        if off >= self.hi:
            return
        if (self.hi - off) < self.bucket_elems and off != 0:
            return
        self._launch(off, self.hi)
        self.hi = off

    def _launch(self, lo, hi):
        G = self.engine.flat.G
        sl = G[lo:hi]
        self.slices.append((lo, hi))
        if self.compress == "bf16":
            from . import ops
            tmp = torch.empty(hi - lo, dtype=torch.bfloat16, device=G.device)
            ops.cast(sl, tmp)
            h = dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append((h, tmp, sl))
        else:
            h = dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append((h, None, sl))

    def finish(self):
        """Wait for every bucket (compute stream waits on the collectives); call before optimizer.step()."""
        if self.world == 1:
            self.hi = None
            return
        if self.hi is not None and self.hi > 0:
            self._launch(0, self.hi)
        for h, tmp, sl in self.handles:
            h.wait()
            if tmp is not None:
                from . import ops
                ops.cast(tmp, sl)
        self.handles, self.hi = [], None
