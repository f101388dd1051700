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


import os as _os

SHARD_ALIGN = 1024        # a rank's shard of a slice is a multiple of this many elements (16-byte pieces in every dtype, whole AdamW blocks)
PACK_SLICE_MIN = int(_os.environ.get("GSTVD_PACK_SLICE_MIN", str(1 << 15)))   # fp32-read pieces at least this long travel as slice copies


class BackwardPipeline(object):
    def __init__(self, engine, optimizer=None, group=None, chunk_elems=40 << 20, compress=None, force_collective=False,
                 keep_grads=False, shard_update=False, direct_bf16=True):
        self.engine, self.opt, self.group = engine, optimizer, group
        self.chunk = chunk_elems
        self.compress = compress
        # shard_update (N > 1, optimizer attached): per slice REDUCE-SCATTER the gradients, run AdamW on this rank's 1/N shard only
        # (fp32 master weights and both moments are current only there), ALL-GATHER the updated bf16 shadow weights (+ the fp32
        # values of the few parameters the forward reads in fp32).  The reference's optimizer runs once, on GPU 0
        # (train_gen.py:326-329); the all-reduce path above runs the full 388 M-parameter AdamW on EVERY rank (1.7 ms of HBM
        # time per step), this one 1/N of it, for the same bytes on the links.  See _run_sharded.
        self.shard_update = bool(shard_update)
        self._plans = {}
        # Staleness of the fp32 masters / moments of the OTHER ranks' shards is derived from DEVICE state, not from a host flag: under
        # hipGraph replay _run_sharded's host code runs once, at capture, while every replay advances the optimizer's device-resident
        # step counter.  `_sharded_seen`: a sharded slice with a non-empty bulk has been issued (eagerly or into a capture) at all;
        # `_synced_step`: the optimizer step count at the last sync_master() (None: never synced).  See the `master_stale` property.
        self._sharded_seen = False
        self._synced_step = None
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
        # graph.SegmentedStep sets this while it captures: a slice's collective (+ update) is then NOT issued into the open capture;
        # run_slice hands it to segmenter.cut(fn), which closes the capture, files `fn` as an eagerly issued item of the replay
        # sequence and opens the next capture (the fall-back for a refused whole-step capture at N > 1: VERDICT r5 item 3a)
        self.segmenter = None
        # bf16 payload without keep_grads: the weight-gradient launch writes the payload of every GEMM weight itself (flat bf16 buffer
        # `Gb`, indexed like G; ops.GemmGroup.flush_direct_bf16) and only the rest of a slice is cast -- 2.8 GB less traffic per step
        self.Gb, self._cast_plans, self.direct_bf16 = None, {}, bool(direct_bf16)
        self._skip_next = self._skipping = False
        self._stale = set()           # flat offsets of the weights whose gradient the LAST backward never stored (fused update)
        engine.pipe = self

    @property
    def master_stale(self):
        """True while sharded optimizer steps have run since the last sync_master(): this rank's fp32 master weights / moments of the
        other ranks' shards are old.  Compares the optimizer's DEVICE step counter (it advances inside a replayed hipGraph; a host
        flag set by _run_sharded would not) with its value at the last sync_master().  Reads the counter: not during a capture."""
        if not (self.shard_update and self.world > 1 and self._sharded_seen and self.opt is not None):
            return False
        if self._synced_step is None:
            return True
        return self.opt._sync_step() != self._synced_step

    def skip_update_once(self):
        """The NEXT backward only produces gradients: no all-reduce, no optimizer step, nothing zeroed -- iteration 0 of
        train_gen.py:326-329 (`if iter_id > 0: optimizer.step(); optimizer.zero_grad()`).  The gradients stay in the flat
        buffer / `.grad`, LOCAL to the rank; the following backward accumulates onto them (leave `.grad` alone in between, as
        the reference does) and that sum is what gets all-reduced and applied, once."""
        self._skip_next = True

    def begin(self):
        """Called by the engine at the start of backward (main stream)."""
        self.hi = self.engine.flat.n_live
        self.slices = []
        self._stale = set()
        self._skipping, self._skip_next = self._skip_next, False
        if self.opt is not None and not self._skipping:
            self.opt.begin_step()

    def fuse_handle(self):
        """Not None when the slice's weight-gradient launch may update its weights itself (see __init__)."""
        if self.collective or self.opt is None or not self.fuse_update or self._skipping:
            return None
        return self.opt.fuse_handle(write_grad=self.keep_grads)

    def bf16_target(self):
        """(G, Gb) when the slice's weight-gradient launch may write its all-reduce payload directly in bf16, else None."""
        flat = self.engine.flat
        if not (self.direct_bf16 and self.collective and self.compress == "bf16" and not self.keep_grads and self.opt is not None
                and not self._skipping and getattr(flat, "G", None) is not None and flat.G.is_cuda and self.use_comm_stream):
            return None
        if self.Gb is None or self.Gb.numel() != flat.G.numel() or self.Gb.device != flat.G.device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("BackwardPipeline: the bf16 gradient buffer has to exist before a hipGraph capture -- run one eager step first")
            self.Gb, self._cast_plans, self._bufs = torch.zeros(flat.G.numel(), dtype=torch.bfloat16, device=flat.G.device), {}, {}
        return flat.G, self.Gb

    def _cast_plan(self, lo, hi, direct):
        """The part of slice [lo, hi) that the weight-gradient launch did NOT write in bf16 (everything but `direct`), as a cached
        range table, and the flat offsets of the parameters that lie in `direct` (their fp32 gradient was never stored)."""
        key = (lo, hi, direct)
        hit = self._cast_plans.get(key)
        if hit is None:
            from . import ops
            rest, pos = [], lo
            for off, n in direct:
                if off < pos or off + n > hi:
                    raise RuntimeError("internal: a directly written gradient block lies outside its slice")
                if off > pos:
                    rest.append((pos, off - pos))
                pos = off + n
            if pos < hi:
                rest.append((pos, hi - pos))
            flat = self.engine.flat
            import bisect
            starts = [o for o, _ in direct]
            stale = set()
            for _, poff in getattr(flat, "items", ()):
                k = bisect.bisect_right(starts, poff) - 1
                if k >= 0 and poff < direct[k][0] + direct[k][1]:
                    stale.add(poff)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("BackwardPipeline: new slice layout during a hipGraph capture -- run one eager step first")
            hit = self._cast_plans[key] = (ops.CastRanges(rest, flat.G.device), frozenset(stale))
        return hit

    def stale_grad_offsets(self):
        """Flat offsets of the tensors whose gradient the backward that just ended did NOT materialise: the weight-gradient
        launch updated them in its epilogue without storing dW (fuse_update without keep_grads).  The engine leaves their
        `.grad` at None instead of a view of stale bytes."""
        return self._stale

    def ready(self, off):
        """True when the completed region [off, hi) should be emitted now.  `chunk_elems` may be a sequence: the k-th slice
        waits for chunk_elems[min(k, last)] elements -- large slices first (their all-reduce has the whole rest of backward
        to hide behind), small ones at the end (the last slice's wgrad + all-reduce + AdamW is the exposed tail)."""
        if isinstance(self.chunk, (list, tuple)):
            need = self.chunk[min(len(self.slices), len(self.chunk) - 1)]
        else:
            need = self.chunk
        return off < self.hi and ((self.hi - off) >= need or off == 0)

    def run_slice(self, lo, hi, fused=(), direct=()):
        """Runs on the auxiliary stream, after the slice's weight-gradient GEMMs and column reductions.

        With a collective the rest of the slice's life -- compression cast, all-reduce, AdamW -- moves to a dedicated
        communication stream that forks from the aux stream here and is joined into the stream backward started on only at
        the end of backward (`end()` returns the event): the aux stream goes straight on to the next slice's weight
        gradients while this slice's message travels.  (Joining it back into the aux stream instead segfaults
        hipStreamEndCapture on this stack: a forked stream may only be joined into the capture's origin stream.)"""
        flat = self.engine.flat
        self.slices.append((lo, hi))
        if hi <= lo or self._skipping:
            self.hi = lo
            return
        if fused and not self.keep_grads:
            self._stale.update(fused)
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
        else:
            plan = None
            if direct or (sl.is_cuda and self.bf16_target() is not None):
                plan = self._cast_plan(lo, hi, tuple(direct))       # (direct may be empty: a slice without GEMM weights)
                self._stale.update(plan[1])
            if self.segmenter is not None:
                if not sl.is_cuda:
                    raise RuntimeError("segmented capture needs device tensors")
                self.segmenter.cut(lambda lo=lo, hi=hi, plan=plan: self._collective_slice(lo, hi, plan))
            else:
                self._collective_slice(lo, hi, plan)
        self.hi = lo

    def _collective_slice(self, lo, hi, plan=None):
        """A finished slice's collective and update, issued on the CURRENT stream's behalf: (device tensors, the default) the work
        forks onto the communication stream and leaves its last event in `tail_event`; host tensors / GSTVD_PIPE_COMM=0 run in line.
        Called from run_slice, or -- segmented replay -- eagerly between two captured segments of the step."""
        flat = self.engine.flat
        sl = flat.G[lo:hi]
        if self.shard_update and self.opt is not None:
            if sl.is_cuda and self.use_comm_stream:
                if self.comm is None:
                    self.comm = torch.cuda.Stream(device=sl.device)
                    self._tick = torch.zeros(1, device=sl.device)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm.wait_event(ev)
                with torch.cuda.stream(self.comm):
                    self._run_sharded(lo, hi, plan)
                    self.tail_event = torch.cuda.Event()
                    self.tail_event.record(self.comm)
            else:
                self._run_sharded(lo, hi)
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
            if plan is not None:
                reduced = self.Gb[lo:hi]                       # the GEMM weights' payload is already there (flush_direct_bf16)
            elif self.compress == "bf16":
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
                    if plan is not None:
                        self._tick.zero_()
                        plan[0].run(flat.G, self.Gb)           # the rest of the slice: biases, LayerNorm, embedding tables
                    else:
                        ops.cast(sl, reduced)
                    dist.all_reduce(reduced, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    self._tick.zero_()
                    dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)
                self._update(lo, hi, reduced)
                self.tail_event = torch.cuda.Event()
                self.tail_event.record(self.comm)

    def segment_join(self):
        """Last item of a segmented replay: the current stream waits for the communication stream's last slice."""
        ev, self.tail_event = self.tail_event, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

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

    # ---- sharded update -----------------------------------------------------------------------------------------------------
    def _plan(self, lo, hi):
        """Static partition of slice [lo, hi) over the ranks (built once per slice, before any hipGraph capture):
          bulk  = [lo, lo + world * S): rank q owns [lo + q S, lo + (q + 1) S), S a multiple of SHARD_ALIGN -- equal shards, so the
                  reduce-scatter and both all-gathers are single in-place-shaped collectives on contiguous memory;
          rest  = [lo + world * S, hi) (< world * SHARD_ALIGN elements): all-reduced and updated by EVERY rank, like the
                  all-reduce path does for everything.
        `pack` = positions inside this rank's shard that the forward reads as fp32 (biases, LayerNorm, embedding tables: everything
        but the GEMM weights, which it reads from the bf16 shadow): their fp32 values travel in a second, small all-gather."""
        key = (lo, hi)
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        flat = self.engine.flat
        dev = flat.G.device
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("BackwardPipeline(shard_update): new slice during a hipGraph capture -- run one eager step first")
        world = self.world
        rank = self.rank_of_plan = dist.get_rank(self.group)
        L = hi - lo
        S = (L // (world * SHARD_ALIGN)) * SHARD_ALIGN
        bulk = world * S
        pay = torch.bfloat16 if self.compress == "bf16" else torch.float32
        pl = dict(S=S, bulk=bulk, rest=L - bulk, a=lo + rank * S, b=lo + (rank + 1) * S)
        pl["red"] = torch.empty(L, dtype=pay, device=dev) if self.compress == "bf16" else None
        pl["shard"] = torch.empty(S, dtype=pay, device=dev) if S else None
        pl["pack"] = None
        shadow = getattr(flat, "S", None)
        if S and shadow is not None:
            ranges = flat.fp32_read_ranges() if hasattr(flat, "fp32_read_ranges") else [(0, flat.n_live)]
            # a rank's packed buffer = its LARGE pieces (an embedding table's share of the shard: plain slice copies) followed by its
            # small ones (biases, LayerNorm: one index gather); `width` = the longest rank's buffer
            BIG = PACK_SLICE_MIN
            per_rank = []
            for q in range(world):
                a, b = lo + q * S, lo + (q + 1) * S
                pieces = [(max(a, x), min(b, y)) for (x, y) in ranges if max(a, x) < min(b, y)]
                big = [pc for pc in pieces if pc[1] - pc[0] >= BIG]
                small = [pc for pc in pieces if pc[1] - pc[0] < BIG]
                sidx = torch.cat([torch.arange(x, y, dtype=torch.int64) for x, y in small]) if small else torch.zeros(0, dtype=torch.int64)
                per_rank.append((big, sidx))
            width = max(sum(y - x for x, y in big) + int(sidx.numel()) for big, sidx in per_rank)
            if width:
                copies, dst, src = [], [], []          # (flat lo, flat hi, offset in `all`) of every rank's large pieces; small: index lists
                for q, (big, sidx) in enumerate(per_rank):
                    off = q * width
                    for x, y in big:
                        copies.append((x, y, off)); off += y - x
                    dst.append(sidx); src.append(off + torch.arange(sidx.numel(), dtype=torch.int64))
                mine_big, mine_small = per_rank[rank]
                n_big = sum(y - x for x, y in mine_big)
                dst, src = torch.cat(dst), torch.cat(src)
                pl["pack"] = dict(width=width, mine_big=mine_big, n_big=n_big, mine_small=mine_small.to(dev), n_small=int(mine_small.numel()),
                                  copies=copies, dst=dst.to(dev) if dst.numel() else None, src=src.to(dev) if src.numel() else None,
                                  buf=torch.zeros(width, dtype=torch.float32, device=dev),
                                  all=torch.empty(world * width, dtype=torch.float32, device=dev))
        self._plans[key] = pl
        return pl

    def _run_sharded(self, lo, hi, plan=None):
        """One slice of the sharded update, on the current stream: [cast] -> reduce-scatter(bulk) + all-reduce(rest) -> AdamW on
        (own shard, rest) -> all-gather(bf16 shadow of the bulk) [-> all-gather(fp32-read parameters)].  Link bytes per rank and
        slice: (N-1)/N x 2 L for the scatter + (N-1)/N x 2 L for the gather = the ring all-reduce's 2 (N-1)/N x 2 L (bf16 payload);
        update cost L / N instead of L.  Without a bf16 shadow (fp32 precision mode) the fp32 weights are gathered instead."""
        flat, opt, pl = self.engine.flat, self.opt, self._plan(lo, hi)
        S, bulk, a, b = pl["S"], pl["bulk"], pl["a"], pl["b"]
        G = flat.G[lo:hi]
        if plan is not None:
            red = self.Gb[lo:hi]                 # GEMM weights' payload written by the weight-gradient launch; the rest cast here
            if getattr(self, "_tick", None) is not None:
                self._tick.zero_()
            plan[0].run(flat.G, self.Gb)
        elif pl["red"] is not None:
            red = pl["red"]
            if G.is_cuda:
                from . import ops
                ops.cast(G, red)
            else:
                red.copy_(G)
        else:
            red = G
            if G.is_cuda and getattr(self, "_tick", None) is not None:
                self._tick.zero_()               # work of our own in front of the first collective on this stream (see run_slice)
        if S:
            dist.reduce_scatter_tensor(pl["shard"], red[:bulk], op=dist.ReduceOp.SUM, group=self.group)
        if pl["rest"]:
            dist.all_reduce(red[bulk:], op=dist.ReduceOp.SUM, group=self.group)
        if red.dtype == torch.bfloat16:
            if self.keep_grads and S:            # `.grad` of the rank's own shard and of the rest (the only parts it has in full)
                flat.G[a:b].copy_(pl["shard"]); flat.G[lo + bulk:hi].copy_(red[bulk:])
            if S:
                opt.apply_range(a, b, grad_bf16=pl["shard"], grad_origin=a)
            if pl["rest"]:
                opt.apply_range(lo + bulk, hi, grad_bf16=red[bulk:], grad_origin=lo + bulk)
        else:
            if S:
                flat.G[a:b].copy_(pl["shard"])   # (the scatter's output cannot alias its input on every backend)
                opt.apply_range(a, b)
            if pl["rest"]:
                opt.apply_range(lo + bulk, hi)
        if S:
            self._sharded_seen = True
            W = flat.S if getattr(flat, "S", None) is not None else flat.P
            dist.all_gather_into_tensor(W[lo:lo + bulk], W[a:b], group=self.group)
            pk = pl["pack"]
            if pk is not None:
                off = 0
                for x, y in pk["mine_big"]:
                    pk["buf"][off:off + y - x].copy_(flat.P[x:y]); off += y - x
                if pk["n_small"]:
                    pk["buf"][off:off + pk["n_small"]].copy_(flat.P.index_select(0, pk["mine_small"]))
                dist.all_gather_into_tensor(pk["all"], pk["buf"], group=self.group)
                mine_lo, mine_hi = self.rank_of_plan * pk["width"], (self.rank_of_plan + 1) * pk["width"]
                for x, y, o in pk["copies"]:
                    if not (mine_lo <= o < mine_hi):          # (this rank's own pieces are already in place)
                        flat.P[x:y].copy_(pk["all"][o:o + y - x])
                if pk["dst"] is not None:
                    flat.P.index_copy_(0, pk["dst"], pk["all"].index_select(0, pk["src"]))

    def sync_master(self):
        """After sharded steps the fp32 master weights and the moments of a slice's bulk are current only on their owner.
        Gathers them (in place, same partition) so that this rank's `state_dict()` / optimizer state / checkpoint is complete
        (train_gen.py:346-357 saves from one process).  A no-op without shard_update; collective: every rank must call it."""
        if not (self.shard_update and self.collective and self.opt is not None):
            return
        flat, opt = self.engine.flat, self.opt
        for (lo, hi) in self.slices:
            pl = self._plans.get((lo, hi))
            if pl is None or not pl["S"]:
                continue
            a, b, bulk = pl["a"], pl["b"], pl["bulk"]
            for buf in (flat.P, opt.m, opt.v):
                dist.all_gather_into_tensor(buf[lo:lo + bulk], buf[a:b].clone(), group=self.group)
        self._synced_step = opt._sync_step()      # (device counter: correct after any number of graph replays)

    def check_master_current(self, what):
        """state_dict() / checkpoint guard: a COLLECTIVE must not hide inside a call that train_gen.py:346-357 makes on one process
        only, so a stale master copy is an error with the way out in the message, never a silent gather or a silent stale file."""
        if self.master_stale:
            from ._lib import GstvdError
            raise GstvdError("%s while the optimizer is sharded over %d ranks: this rank holds current fp32 master weights / moments only "
                             "for its own shards.  Call pipe.sync_master() on EVERY rank first (one all-gather per slice), then save "
                             "from whichever rank you like." % (what, self.world))

    def end(self):
        """End of backward.  Returns the event the caller's stream must wait on (the communication stream's last work) or None."""
        if self.opt is not None and not self._skipping:
            self.opt._applied_in_backward = True
        ev, self.tail_event = self.tail_event, None
        return ev
