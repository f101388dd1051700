"""Backward pipeline: finalise gradients slice by slice on a third HIP stream while backward is still running.

Backward completes the flat gradient buffer from its end (LM head, decoder) to its start (embeddings).  Whenever
a contiguous slice [lo, hi) of at least `chunk_elems` elements is complete the engine hands it to this object,
which -- on the engine's auxiliary stream, i.e. concurrently with the remaining backward kernels -- runs

    1. the deferred weight-gradient GEMMs of the slice (one grouped launch) and its column reductions,
    2. for N > 1: a RCCL sum all-reduce of G[lo:hi] (one large contiguous collective per slice; xGMI links are
       point to point, so few large messages beat many small ones),
    3. the fused AdamW update of P[lo:hi] (+ bf16 shadow weights), with the 1/N scale folded in;
    at N > 1 steps 2-3 ride on a dedicated communication stream so that the aux stream can start the next slice.

It is the ONE gradient-synchronisation path of this package (round 1 also carried a second, hook-based bucketed all-reduce,
`dp.GradSync`, that duplicated the slicing logic and was not on the bench path; it is gone).  Sizing: xGMI is point to point
(7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound, so slices are few and large (graded: 128, 96, 96, 32,
16 Mi elements at N>1 -- the first, largest one has the whole encoder backward to hide behind, the last ones keep the exposed
tail short) instead of hundreds of small buckets.

This is the MI355X-native replacement of nn.DataParallel's per-step parameter broadcast + gradient reduce-add
(train_gen.py:295,324) and of the serial optimizer.step() (train_gen.py:326-329).  loss = mean over ranks of the
per-rank token mean, exactly DataParallel's gather + .mean() (train_gen.py:134-135).
"""
import torch
import torch.distributed as dist


class BackwardPipeline(object):
    def __init__(self, engine, optimizer=None, group=None, chunk_elems=40 << 20, compress=None, force_collective=False,
                 keep_grads=False):
        self.engine, self.opt, self.group = engine, optimizer, group
        self.chunk = chunk_elems
        self.compress = compress
        # compress="bf16": the slice is cast to bf16, all-reduced, and (with an optimizer attached) consumed by AdamW
        # straight from the bf16 copy; the fp32 buffer G / `.grad` then keeps the LOCAL gradients unless keep_grads
        self.keep_grads = keep_grads
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.hi = None
        self.slices = []
        self._bufs, self.comm, self.tail_event = {}, None, None
        import os
        self.use_comm_stream = os.environ.get("GSTVD_PIPE_COMM", "1") != "0"
        # N = 1: a slice's AdamW (HBM-bound) runs on a stream of its own, beside the NEXT slice's weight-gradient GEMMs
        # (MFMA-bound) and the rest of backward.  Zero-sum with round 2's two large slices; with the decoder's gradients finished
        # in small slices during its own backward it is worth 0.35 ms per step (tools/chunk_sweep.sh: 13.65 vs 14.00 ms)
        self.update_stream = os.environ.get("GSTVD_PIPE_UPDATE_STREAM", "1") != "0"
        # N = 1: nothing stands between a weight gradient and its update, so the grouped weight-gradient launch applies AdamW in
        # its epilogue (gstvd_gemm_grouped_adamw: 26 bytes per weight instead of 4 + 30, streamed by the tile's idle producer
        # waves under the other workgroups' K-loops) and the AdamW pass shrinks to the ~5 % of the parameters that are not GEMM
        # weights.  Bit-identical to the two launches.  With keep_grads the launch also stores dW for `.grad`.
        self.fuse_update = os.environ.get("GSTVD_FUSE_UPDATE", "1") != "0"
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        engine.pipe = self

    def begin(self):
        """Called by the engine at the start of backward (main stream)."""
        self.hi = self.engine.flat.n_live
        self.slices = []
        if self.opt is not None:
            self.opt.begin_step()

    def fuse_handle(self):
        """Not None when the slice's weight-gradient launch may update its weights itself (see __init__)."""
        if self.collective or self.opt is None or not self.fuse_update:
            return None
        return self.opt.fuse_handle(write_grad=self.keep_grads)

    def ready(self, off):
        """True when the completed region [off, hi) should be emitted now.  `chunk_elems` may be a sequence: the k-th slice
        waits for chunk_elems[min(k, last)] elements -- large slices first (their all-reduce has the whole rest of backward
        to hide behind), small ones at the end (the last slice's wgrad + all-reduce + AdamW is the exposed tail)."""
        if isinstance(self.chunk, (list, tuple)):
            need = self.chunk[min(len(self.slices), len(self.chunk) - 1)]
        else:
            need = self.chunk
        return off < self.hi and ((self.hi - off) >= need or off == 0)

    def run_slice(self, lo, hi, fused=()):
        """Runs on the auxiliary stream, after the slice's weight-gradient GEMMs and column reductions.

        With a collective the rest of the slice's life -- compression cast, all-reduce, AdamW -- moves to a dedicated
        communication stream that forks from the aux stream here and is joined into the stream backward started on only at
        the end of backward (`end()` returns the event): the aux stream goes straight on to the next slice's weight
        gradients while this slice's message travels.  (Joining it back into the aux stream instead segfaults
        hipStreamEndCapture on this stack: a forked stream may only be joined into the capture's origin stream.)"""
        flat = self.engine.flat
        self.slices.append((lo, hi))
        if hi <= lo:
            self.hi = lo
            return
        sl = flat.G[lo:hi]
        if not self.collective:
            if self.update_stream and sl.is_cuda and self.opt is not None:
                # N = 1: the slice's AdamW (HBM-bound) moves to its own stream so that it runs beside the NEXT slice's
                # weight-gradient GEMMs (MFMA-bound) and the rest of backward instead of in front of them on the aux stream
                if self.comm is None:
                    self.comm = torch.cuda.Stream(device=sl.device)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm.wait_event(ev)
                with torch.cuda.stream(self.comm):
                    self._update(lo, hi, None, fused)
                    self.tail_event = torch.cuda.Event()
                    self.tail_event.record(self.comm)
            else:
                self._update(lo, hi, None, fused)
        elif not sl.is_cuda or not self.use_comm_stream:       # host tensors (gloo tests) / in-line variant
            reduced = None
            if self.compress == "bf16":
                if sl.is_cuda:
                    from . import ops
                    reduced = torch.empty(hi - lo, dtype=torch.bfloat16, device=sl.device)
                    ops.cast(sl, reduced)
                else:
                    reduced = sl.to(torch.bfloat16)
                dist.all_reduce(reduced, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)
            self._update(lo, hi, reduced)
        else:
            from . import ops
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=sl.device)
                self._tick = torch.zeros(1, device=sl.device)
            reduced = None
            if self.compress == "bf16":
                reduced = self._bufs.get((lo, hi))             # persistent: no allocator traffic across streams
                if reduced is None:
                    reduced = self._bufs[(lo, hi)] = torch.empty(hi - lo, dtype=torch.bfloat16, device=sl.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                # work of our own precedes the collective on this stream (the compression cast, or a 4-byte memset): a
                # stream that has only waited on a captured event does not report itself as capturing yet
                if reduced is not None:
                    ops.cast(sl, reduced)
                    dist.all_reduce(reduced, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    self._tick.zero_()
                    dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)
                self._update(lo, hi, reduced)
                self.tail_event = torch.cuda.Event()
                self.tail_event.record(self.comm)
        self.hi = lo

    def _update(self, lo, hi, reduced, fused=()):
        if reduced is not None and (self.opt is None or self.keep_grads):
            if reduced.is_cuda:
                from . import ops
                ops.cast(reduced, self.engine.flat.G[lo:hi])
            else:
                self.engine.flat.G[lo:hi].copy_(reduced)
            reduced = None
        if self.opt is not None:
            self.opt.apply_range(lo, hi, grad_bf16=reduced, fused=fused)

    def end(self):
        """End of backward.  Returns the event the caller's stream must wait on (the communication stream's last work) or None."""
        if self.opt is not None:
            self.opt._applied_in_backward = True
        ev, self.tail_event = self.tail_event, None
        return ev
