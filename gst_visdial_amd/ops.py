"""Thin tensor-level wrappers over the C ABI (include/gstvd_hip.h).

torch is used only as the owner of device memory and of the HIP stream; every function here launches
a hand-written gfx950 kernel on `torch.cuda.current_stream()` and raises if the library is missing,
a tensor is not on the GPU, or the kernel returns a non-zero status.  No fallbacks.
"""
import os
import ctypes as C

import torch

from . import _lib as L
from ._lib import F32, BF16, EPI_BIAS, EPI_ADD, EPI_GELU, EPI_DGELU, EPI_DROPOUT, LN_RESID, LN_EMBED, LN_IMAGE  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t):
    return _DT[t.dtype]


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.GstvdError("gst_visdial_amd ops need GPU tensors (got %s); there is no CPU path" % t.device)
    return t.data_ptr()


# The handle of the current HIP stream is needed by every library call (~330 per eagerly issued step).  The public route --
# torch.cuda.current_stream().cuda_stream -- builds a Stream object and resolves the device index through four Python layers:
# 2.55 us per call, 0.85 ms of host time per step (tools/host_profile.py, profiles/r06_host_profile.txt).  torch's own raw
# accessor returns the same handle in ~0.2 us; the engine tells this module which device it drives (one per process).
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_DEV_INDEX = None


def set_device(device):
    global _DEV_INDEX
    _DEV_INDEX = device.index if device.index is not None else torch.cuda.current_device()


def _stream():
    if _RAW_STREAM is not None and _DEV_INDEX is not None:
        return _RAW_STREAM(_DEV_INDEX)
    return torch.cuda.current_stream().cuda_stream


GEMM256_MIN_TILES = int(__import__("os").environ.get("GSTVD_GEMM256_MIN_TILES", "120"))


class Profiler(object):
    """Optional per-launch timing with HIP events on the launch stream (bench.py's kernel breakdown).
    Off by default; `with ops.Profiler() as p:` records (tag, flops, bytes, start, end) per wrapped call."""
    active = None
    scope = None          # label the engine sets around a block (e.g. "coattn"): carried by every record made inside it

    def __init__(self):
        self.records = []

    def __enter__(self):
        Profiler.active = self
        return self

    def __exit__(self, *a):
        Profiler.active = None

    def summary(self, by_shape=False, scope=None):
        """Aggregate by tag; `scope`: only the records made inside that engine block."""
        torch.cuda.synchronize()
        agg = {}
        for tag, flops, nbytes, e0, e1, detail, sc in self.records:
            if scope is not None and sc != scope:
                continue
            if by_shape and detail is not None:
                tag = "%s %s" % (tag, "x".join(str(d) for d in detail))
            a = agg.setdefault(tag, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


def _prof_begin():
    if Profiler.active is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, tag, flops=0.0, nbytes=0.0, detail=None):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    Profiler.active.records.append((tag, flops, nbytes, e0, e1, detail, Profiler.scope))


def gemm_tag(dtype_in, a_km, b_km, M, N, batch):
    """Name of the kernel instantiation gstvd_gemm dispatches to (same rule as launch_layout in csrc/gemm.hip)."""
    big = ((M + 127) // 128) * ((N + 127) // 128) * batch
    tile = 128 if (M >= 256 and N >= 128 and big >= 96) else 64
    if dtype_in == BF16 and M >= 256 and N >= 256 and ((M + 255) // 256) * ((N + 255) // 256) * batch >= GEMM256_MIN_TILES:
        tile = 256
    lay = {(0, 0): "nt", (0, 1): "nn", (1, 1): "tn", (1, 0): "tt"}[(int(a_km), int(b_km))]
    return "gemm_%s_%s_%d" % ("bf16" if dtype_in == BF16 else "f32", lay, tile)


_KNAME = {}


def gemm_kernel_symbol(d, splits=1):
    """(Mangled) symbol of the device kernel the library launches for descriptor `d` -- asked of the library's own dispatch
    (gstvd_gemm_kernel_name), cached by everything the dispatch looks at."""
    key = (d.M, d.N, d.K, d.batch, d.dtype_in, d.dtype_out, d.a_kmajor, d.b_kmajor, splits, d.ldc % 8, d.epilogue & ~0)
    name = _KNAME.get(key)
    if name is None:
        buf = C.create_string_buffer(512)
        L.check("gstvd_gemm_kernel_name", L.load().gstvd_gemm_kernel_name(C.byref(d), splits, buf, 512))
        name = _KNAME[key] = buf.value.decode()
    return name


def rank_seed(seed, rank):
    """Dropout seed of data-parallel rank `rank`: every nn.DataParallel replica of the reference draws its masks from its own
    device's generator (train_gen.py:295), so ranks must not share a mask stream.  Rank 0 keeps `seed`."""
    return (int(seed) + int(rank) * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF


class Rng:
    """Device-resident (seed, offset) pair read by every dropout site; graph-capture safe."""

    def __init__(self, device, seed=0):
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)

    def advance(self):
        lib = L.load()
        L.check("gstvd_rng_advance", lib.gstvd_rng_advance(self.state.data_ptr(), _stream()))

    def ptr(self):
        return self.state.data_ptr()


def gemm(A, B, C_out, M, N, K, *, a_km=False, b_km=False, bias=None, addend=None, aux=None, epi=0,
         alpha=1.0, lda=None, ldb=None, ldc=None, ldadd=None, ldaux=None, batch=1, sA=0, sB=0, sC=0, sAdd=0, sAux=0,
         drop_p=0.0, site=0, rng=None):
    """C[M,N] = epi(alpha * A(m,k) B(n,k)); see gstvd_gemm in include/gstvd_hip.h.  Leading dimensions
    default to the tensors' row strides."""
    lib = L.load()
    d = L.GemmDesc()
    d.A, d.B, d.C = _p(A), _p(B), _p(C_out)
    d.bias, d.addend, d.aux = _p(bias), _p(addend), _p(aux)
    d.M, d.N, d.K = M, N, K
    d.lda = A.stride(-2) if lda is None else lda
    d.ldb = B.stride(-2) if ldb is None else ldb
    d.ldc = C_out.stride(-2) if ldc is None else ldc
    d.ldadd = (addend.stride(-2) if addend is not None else 0) if ldadd is None else ldadd
    d.ldaux = (aux.stride(-2) if aux is not None else 0) if ldaux is None else ldaux
    d.batch, d.sA, d.sB, d.sC, d.sAdd, d.sAux = batch, sA, sB, sC, sAdd, sAux
    d.dtype_in, d.dtype_out = dt(A), dt(C_out)
    d.a_kmajor, d.b_kmajor = int(a_km), int(b_km)
    if bias is not None:
        epi |= EPI_BIAS
    if addend is not None:
        epi |= EPI_ADD
    if drop_p > 0:
        epi |= EPI_DROPOUT
    d.epilogue = epi
    d.alpha, d.dropout_p, d.site = alpha, drop_p, site
    d.rng = rng.ptr() if (rng is not None and drop_p > 0) else None
    e0 = _prof_begin()
    splits = splitk_plan(d.dtype_in, M, N, K, batch, a_km, b_km)
    if splits > 1:
        ws = _splitk_scratch(A.device)
        L.check("gstvd_gemm_splitk", lib.gstvd_gemm_splitk(C.byref(d), splits, ws.data_ptr(), ws.numel(), _stream()))
    else:
        L.check("gstvd_gemm", lib.gstvd_gemm(C.byref(d), _stream()))
    # profiling: the record is keyed by the symbol the library really launched (one per template instantiation)
    tag = ("gemm:" + gemm_kernel_symbol(d, splits)) if e0 is not None else None
    _prof_end(e0, tag, 2.0 * M * N * K * batch,
              float(batch) * ((M * K + N * K) * A.element_size() + M * N * C_out.element_size()), (M, N, K, batch))
    return C_out


def gemv_ln(A, B, C_out, M, N, K, gamma, beta, eps, y_out=None, bias=None, addend=None, aux=None, epi=0):
    """C[M <= 16, N] = epi(LayerNorm(A[M, K]; gamma, beta, eps) . B[N, K]^T), bf16 operands (gstvd_gemv_ln): the decode step's
    LayerNorm -> Linear pairs as one launch; y_out [M, K] bf16 (optional) receives the normalised rows."""
    lib = L.load()
    d = L.GemmDesc()
    d.A, d.B, d.C = _p(A), _p(B), _p(C_out)
    d.bias, d.addend, d.aux = _p(bias), _p(addend), _p(aux)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = A.stride(-2), B.stride(-2), C_out.stride(-2)
    d.ldadd = addend.stride(-2) if addend is not None else 0
    d.ldaux = aux.stride(-2) if aux is not None else 0
    d.batch, d.dtype_in, d.dtype_out, d.alpha = 1, dt(A), dt(C_out), 1.0
    if bias is not None:
        epi |= EPI_BIAS
    if addend is not None:
        epi |= EPI_ADD
    d.epilogue = epi
    e0 = _prof_begin()
    L.check("gstvd_gemv_ln", lib.gstvd_gemv_ln(C.byref(d), _p(gamma), _p(beta), float(eps), _p(y_out),
                                              y_out.stride(-2) if y_out is not None else 0, _stream()))
    _prof_end(e0, "gemv_ln", 2.0 * M * N * K, float((M * K + N * K) * 2 + M * N * C_out.element_size()), (M, N, K, 1))
    return C_out


SPLITK = int(os.environ.get("GSTVD_GEMM_SPLITK", "1"))
_SPLITK_WS = {}


def splitk_plan(dtype_in, M, N, K, batch, a_km, b_km):
    """Number of K splits for the 64x64-tile kernel (1 = plain launch).  Skinny and deep only: the tiles must leave
    most of the 256 CUs idle and each split must still run a ring's worth of K-tiles."""
    if not SPLITK or dtype_in != BF16 or batch != 1 or (a_km and not b_km):
        return 1
    if M <= 16 and not a_km and not b_km:
        return 1                                    # the skinny kernel of csrc/gemv.hip takes it (one launch, K split over waves)
    if M >= 256 and N >= 128 and ((M + 127) // 128) * ((N + 127) // 128) >= 96:
        return 1                                    # the 128 / 256 tile kernels take it
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    nkt = (K + 63) // 64
    s = min(256 // max(tiles, 1), nkt // 10, 8)
    return s if s >= 2 else 1


def _splitk_scratch(device):
    """Per-stream split-K scratch (launches on one stream never overlap): counters + 256 tile slots, zeroed once."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.zeros(4096 + 256 * 64 * 64 * 4, dtype=torch.uint8, device=device)
        _SPLITK_WS[key] = ws
    return ws


# Tile placement of the grouped launches (xcd_block_map): 3 = per-XCD queues in ROUNDS of an XCD's 32 CUs with short-K fillers
# for the fused weight-gradient + AdamW launch, plain per-XCD queues of whole units (mode 1) for the unfused launch of the N > 1
# path.  0 = the library's own chunked order (rounds 1-4; tests compare against it).  Round 5's other candidates (staggered
# lead-in, 30-tile pieces) were measured and are gone: profiles/r05_group_order_ab.txt, r05_group_rounds_ab.txt.
GROUP_ORDER = 3
N_XCD = 8
XCD_CUS = 32                                # CUs of one XCD = workgroups of the 139 KB-LDS grouped kernels it runs at a time
GROUP_ORDER_MIN_TILES = 4 * N_XCD * 32      # a few rounds of the chip at least: below that the order is moot (tests lower it)


def xcd_block_map(shapes, tile, fused_epilogue, mode=1, unit_tiles=40):
    """Placement of a grouped launch's tiles (gstvd_gemm_grouped*'s block_map_dev): -> list of tile ids, one per workgroup, -1 =
    idle; entries b, b + 8, b + 16, ... are the queue of ONE XCD (workgroups are dealt to the eight XCDs round-robin).

    Why: a 256 x 256 weight-gradient tile streams two [K, 256] operand panels (K = the batch rows: 2 MB each at K = 4096) and the
    36 tiles of a [3072, 768] problem share 15 panels between them.  They only meet in an XCD's 4 MB L2 when they run on that XCD
    AT THE SAME TIME: the ~32 tiles an XCD has in flight must belong to the same problem(s) and walk K in step.  Rounds 1-4 dealt
    chunks of 8 tiles round-robin, so an XCD's 32 tiles were four chunks from four places of a table that mixes K = 400 / 592 /
    4096 problems: equal-K tiles drifted apart within a round or two and the launch fetched its operands 3.5x (r04_pmc_traffic).
    Here: (1) a problem (or, for the few large ones, a contiguous range of <= `unit_tiles` of its tiles) is a UNIT that goes to
    one XCD whole; (2) units are sorted into classes of equal K -- equal running time -- long K first, and dealt to the XCD
    with the least work so far (LPT), so every queue is long-K units back to back, then the short ones: tiles that start
    together finish together and the next 32 start together again; (3) mode 3: the queue is cut into ROUNDS (below).
    `shapes`: [(M, N, K)] per problem in table order; tile ids follow tile_off (problem after problem)."""
    T = tile
    if mode not in (1, 3):
        raise ValueError("xcd_block_map: mode 1 (queues of whole units) or 3 (rounds with short-K fillers)")
    rounds = mode == 3
    if rounds:
        unit_tiles = min(unit_tiles, XCD_CUS - 5)    # 27: a unit must fit one round of an XCD's CUs with room for a few fillers
    units = []                                   # (K, first tile id, number of tiles)
    gid = 0
    for (M, N, K) in shapes:
        nt = ((M + T - 1) // T) * ((N + T - 1) // T)
        parts = max(1, (nt + unit_tiles - 1) // unit_tiles)
        sizes = [nt // parts + (1 if i < nt % parts else 0) for i in range(parts)]
        t0 = gid
        for n in sizes:
            units.append((K, t0, n))
            t0 += n
        gid += nt
    epi = 30.0 if fused_epilogue else 9.0        # us per tile outside the K loop (fused: 1.7 MB of optimizer state per tile)

    def cost(K, n):
        return n * (epi + 5.0 + 0.76 * ((K + 31) // 32))

    kmax = max(u[0] for u in units)
    long_u = sorted([u for u in units if 2 * u[0] > kmax], key=lambda u: (-u[0], -u[2], u[1]))
    short_u = sorted([u for u in units if 2 * u[0] <= kmax], key=lambda u: (-u[0], -u[2], u[1]))
    load = [0.0] * N_XCD
    ql, qs = [[] for _ in range(N_XCD)], [[] for _ in range(N_XCD)]
    for group, dst in ((long_u, ql), (short_u, qs)):
        for u in group:
            x = min(range(N_XCD), key=lambda i: (load[i], i))
            dst[x].append(u)
            load[x] += cost(u[0], u[2])
    queues = []
    for x in range(N_XCD):
        q = []
        if rounds and ql[x]:
            # ROUNDS (round 5, second half): an XCD runs XCD_CUS = 32 of these workgroups at a time (139 KB of LDS: one per CU), and
            # equal-K tiles that start together finish together.  With the long units simply back to back, the window of 32 running
            # tiles slides across unit boundaries: the first few tiles of a 27- or 36-tile unit run a round earlier than the rest and
            # stream their operand panels alone.  Here the long units are packed into rounds of <= 32 tiles (first fit, largest
            # first), and the CUs a round leaves over are kept busy with SHORT-K tiles -- as many as fit into a long tile's time --
            # so that the next round's long tiles find all their CUs free at the same moment.
            shorts = [t for (_, t0, n) in qs[x] for t in range(t0, t0 + n)]
            t_long = cost(ql[x][0][0], 1)
            t_short = cost(qs[x][0][0], 1) if qs[x] else t_long
            per_slot = max(1, int(t_long / t_short))
            bins = []
            for u in sorted(ql[x], key=lambda u: (-u[2], -u[0], u[1])):
                for b in bins:
                    if b[0] + u[2] <= XCD_CUS:
                        b[0] += u[2]; b[1].append(u); break
                else:
                    bins.append([u[2], [u]])
            si = 0
            for nl, us in bins:
                for (_, t0, n) in us:
                    q.extend(range(t0, t0 + n))
                nf = min((XCD_CUS - nl) * per_slot, len(shorts) - si)
                q.extend(shorts[si:si + nf])
                si += nf
            q.extend(shorts[si:])
        else:
            for (_, t0, n) in ql[x] + qs[x]:
                q.extend(range(t0, t0 + n))
        queues.append(q)
    depth = max(len(q) for q in queues)
    out = [-1] * (depth * N_XCD)
    for x, q in enumerate(queues):
        out[x:x + len(q) * N_XCD:N_XCD] = q
    return out


class GemmGroup(object):
    """Deferred GEMMs that share dtypes / operand layouts, executed as ONE grouped launch (gstvd_gemm_grouped).
    The engine queues every weight-gradient GEMM of a backward pass here.  Device tables are cached by content
    (arena addresses are static from step to step), so steady state is a single launch with no host->device copy."""

    def __init__(self, device, a_km=True, b_km=True):
        self.device, self.a_km, self.b_km = device, a_km, b_km
        self.items, self.cache, self.keep = [], {}, []
        self.dtype_in = self.dtype_out = None
        self._caps = None

    def colsum_capable(self, A):
        """True when a grouped launch of this group can also produce bias[m] = sum_k A[k][m] (GSTVD_EPI_COLSUM): bf16 operands
        (the grouped kernel), k-major A, and the library says its grouped launch honours the flag."""
        if not (self.a_km and A.dtype == torch.bfloat16):
            return False
        if self._caps is None:
            self._caps = int(L.load().gstvd_gemm_group_caps())
        return bool(self._caps & 1)

    def add(self, A, B, C_out, M, N, K, accumulate, colsum_out=None, colsum_acc=False):
        di, do = dt(A), dt(C_out)
        if self.items and (di, do) != (self.dtype_in, self.dtype_out):
            self.flush()
        self.dtype_in, self.dtype_out = di, do
        self.keep.append((A, B, C_out, colsum_out))      # keep the operands alive until the launch
        cs = (_p(colsum_out), int(bool(colsum_acc))) if colsum_out is not None else (0, 0)
        self.items.append((_p(A), _p(B), _p(C_out), M, N, K, A.stride(-2), B.stride(-2), C_out.stride(-2), int(bool(accumulate))) + cs)

    def reset(self):
        self.items, self.keep = [], []

    def pending_into(self, C_out):
        """True when a queued problem writes the block that starts at C_out's address.  A NON-GEMM writer into the same gradient
        slot (the embedding scatter-add when the LM head is tied to the decoder's own word embedding) asks this before it
        accumulates: the queued GEMM would otherwise run later and overwrite what was added."""
        ptr = _p(C_out)
        return any(it[2] == ptr for it in self.items)

    @staticmethod
    def _overlapping(items):
        """Indices of queued problems whose C block shares bytes with another queued problem's (sorted sweep over [lo, hi))."""
        spans = sorted((it[2], it[2] + 4 * ((it[3] - 1) * it[8] + it[4]), i) for i, it in enumerate(items))
        bad, reach, owner = set(), -1, -1
        for lo, hi, i in spans:
            if lo < reach:
                bad.add(i); bad.add(owner)
            if hi > reach:
                reach, owner = hi, i
        return bad

    def flush_direct_bf16(self, G, Gb):
        """The N > 1 path with a bf16 gradient payload: every queued weight gradient that is the ONLY contribution to its weight
        this step is written by the launch ITSELF in bf16 into `Gb` (the flat bf16 buffer the slice's collective sends, indexed
        like the fp32 gradient buffer `G`) instead of in fp32 into G -- the fp32 store (4 B per weight) and the cast pass that would
        read it back (4 B) never happen; the rounding is the cast's (round to nearest even of the fp32 accumulator), so the payload
        is bit-identical.  The other problems (accumulating, or sharing their block with another writer: the tied LM head) stay
        fp32.  Two launches instead of one.  Returns the flat (offset, numel) ranges written in bf16, sorted."""
        if not self.items or self.dtype_in != BF16 or self.dtype_out != F32 or not (self.a_km and self.b_km):
            self.flush()
            return ()
        base, nG = G.data_ptr(), G.numel()
        shared = self._overlapping(self.items)
        direct, rest, keep_d, keep_r, ranges = [], [], [], [], []
        for i, (it, kp) in enumerate(zip(self.items, self.keep)):
            (a, b, c, M, N, K, lda, ldb, ldc, acc) = it[:10]
            off = (c - base) // 4
            if acc or i in shared or ldc != N or c < base or off + M * N > nG or (c - base) % 4 or off % 8:
                rest.append(it); keep_r.append(kp)
            else:
                direct.append(it[:2] + (Gb.data_ptr() + 2 * off,) + it[3:]); keep_d.append(kp)
                ranges.append((off, M * N))
        self.items, self.keep = rest, keep_r
        self.flush()
        if direct:
            self.items, self.keep, self.dtype_out = direct, keep_d, BF16
            try:
                self.flush()
            finally:
                self.dtype_out = F32
        return tuple(sorted(ranges))

    def flush(self, fuse=None):
        """Launch the queued problems.  `fuse` (optim.FusedAdamW.fuse_handle(), single-GPU training only): every queued weight
        gradient that is the ONLY contribution to its weight this step gets GSTVD_EPI_ADAMW -- the launch updates the weight in
        its epilogue (gstvd_gemm_grouped_adamw) instead of storing dW.  Returns the flat offsets of the weights updated that way
        (a tuple, empty without fusion): the caller's remainder pass must leave them alone."""
        if not self.items:
            return ()
        keep, self.keep = self.keep, []      # released when this call returns (after the launch is enqueued)
        fused = ()
        if fuse is not None and self.dtype_in == BF16 and self.dtype_out == F32 and self.a_km and self.b_km:
            items, fl = [], []
            # a weight is updated in the launch only when this problem is the ONLY contribution to it this step: not when it
            # accumulates, and not when another queued problem (accumulating or not) writes into the same block
            shared = self._overlapping(self.items)
            for i, it in enumerate(self.items):
                (a, b, c, M, N, K, lda, ldb, ldc, acc, cs_ptr, cs_acc) = it[:12]
                hp_addr, offs = (0, ()) if (acc or i in shared) else fuse.cover(c, M, N, ldc)
                fl.extend(offs)
                items.append(it[:12] + (hp_addr,))
            if fl:
                self.items, fused = items, tuple(fl)
            else:
                fuse = None
        else:
            fuse = None
        if self.dtype_in != BF16:            # fp32 parity mode: plain launches
            lib = L.load()
            for (a, b, c, M, N, K, lda, ldb, ldc, acc, cs_ptr, cs_acc) in [it[:12] for it in self.items]:
                if cs_ptr:
                    raise L.GstvdError("column sums ride on the grouped bf16 launch only")
                d = L.GemmDesc()
                d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.batch = a, b, c, M, N, K, lda, ldb, ldc, 1
                d.dtype_in, d.dtype_out, d.a_kmajor, d.b_kmajor, d.alpha = self.dtype_in, self.dtype_out, int(self.a_km), int(self.b_km), 1.0
                if acc:
                    d.addend, d.ldadd, d.epilogue = c, ldc, EPI_ADD
                L.check("gstvd_gemm", lib.gstvd_gemm(C.byref(d), _stream()))
            self.items = []
            return ()
        key = tuple(self.items)
        hit = self.cache.get(key)
        if hit is None:
            arr = (L.GemmDesc * len(key))()
            offs, tiles, flops, nbytes = [], 0, 0.0, 0.0
            esz_in, esz_out = (2 if self.dtype_in == BF16 else 4), (2 if self.dtype_out == BF16 else 4)
            T = int(L.load().gstvd_gemm_group_tile())
            for d, it in zip(arr, key):
                (a, b, c, M, N, K, lda, ldb, ldc, acc, cs_ptr, cs_acc) = it[:12]
                d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.batch = a, b, c, M, N, K, lda, ldb, ldc, 1
                d.dtype_in, d.dtype_out, d.a_kmajor, d.b_kmajor, d.alpha = self.dtype_in, self.dtype_out, int(self.a_km), int(self.b_km), 1.0
                if acc:
                    d.addend, d.ldadd, d.epilogue = c, ldc, EPI_ADD
                elif len(it) > 12 and it[12]:   # the weight's (lr, wd) pair: updated in the launch's epilogue
                    d.addend, d.epilogue = it[12], L.EPI_ADAMW
                if cs_ptr:                      # bias[m] (+)= sum_k A[k][m] out of the same launch
                    d.bias = cs_ptr
                    d.epilogue |= L.EPI_COLSUM | (L.EPI_COLSUM_ACC if cs_acc else 0)
                offs.append(tiles)
                tiles += ((M + T - 1) // T) * ((N + T - 1) // T)
                flops += 2.0 * M * N * K
                if len(it) > 12 and it[12]:
                    nbytes += esz_in * (M * K + K * N) + M * N * 26.0       # param / m / v in and out + bf16 shadow, no dW
                else:
                    nbytes += esz_in * (M * K + K * N) + esz_out * M * N * (2 if acc else 1)
            tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            off = torch.tensor(offs, dtype=torch.int32).to(self.device)
            bmap = None
            if GROUP_ORDER and tiles >= GROUP_ORDER_MIN_TILES:
                # (rounds with fillers pay in the FUSED launch, whose tiles carry 50 us epilogues and which runs near the fabric's
                # rate: 13.50 -> 13.15 GB, -2.6 %; the plain launch of the N > 1 path is not bound by bytes and fetches less with
                # whole 36-tile units: 3.44 vs 3.86 GB stand-alone, profiles/r05_group_rounds_ab.txt)
                mode = 3 if fuse is not None else 1
                bm_host = xcd_block_map([(it[3], it[4], it[5]) for it in key], T, fuse is not None, mode)
                # the C side's contract -- every tile id exactly once, the rest idle (-1) -- is checked HERE, where the map is
                # made (once per cached table): an id out of range would compute a tile past its problem's extent
                if sorted(t for t in bm_host if t >= 0) != list(range(tiles)) or any(t < -1 for t in bm_host):
                    raise L.GstvdError("internal: block map is not a placement of the launch's %d tiles" % tiles)
                bmap = torch.tensor(bm_host, dtype=torch.int32).to(self.device)
            hit = (tab, off, len(key), tiles, flops, nbytes, bmap)
            if len(self.cache) > 64:
                self.cache.clear()
            self.cache[key] = hit
        tab, off, n, tiles, flops, nbytes, bmap = hit
        bm_ptr, bm_n = (bmap.data_ptr(), bmap.numel()) if bmap is not None else (None, 0)
        lib = L.load()
        e0 = _prof_begin()
        if fuse is not None:
            L.check("gstvd_gemm_grouped_adamw", lib.gstvd_gemm_grouped_adamw(tab.data_ptr(), off.data_ptr(), n, tiles, C.byref(fuse.desc()),
                                                                             bm_ptr, bm_n, _stream()))
            if e0 is not None:
                name = _KNAME.get("grouped_adamw")
                if name is None:
                    buf = C.create_string_buffer(512)
                    L.check("gstvd_gemm_grouped_adamw_kernel_name", lib.gstvd_gemm_grouped_adamw_kernel_name(buf, 512))
                    name = _KNAME["grouped_adamw"] = buf.value.decode()
                _prof_end(e0, "gemm:" + name, flops, nbytes, (n, tiles))
            self.items = []
            return fused
        L.check("gstvd_gemm_grouped", lib.gstvd_gemm_grouped(tab.data_ptr(), off.data_ptr(), n, tiles, self.dtype_in, self.dtype_out,
                                                             int(self.a_km), int(self.b_km), bm_ptr, bm_n, _stream()))
        if e0 is not None:
            key = ("grouped", self.dtype_in, self.dtype_out, self.a_km, self.b_km)
            name = _KNAME.get(key)
            if name is None:
                buf = C.create_string_buffer(512)
                L.check("gstvd_gemm_grouped_kernel_name", lib.gstvd_gemm_grouped_kernel_name(self.dtype_in, self.dtype_out, int(self.a_km),
                                                                                           int(self.b_km), buf, 512))
                name = _KNAME[key] = buf.value.decode()
            _prof_end(e0, "gemm:" + name, flops, nbytes, (n, tiles))
        self.items = []
        return ()


def _ln_desc(mode, dtype, M, H, gamma, beta, mean, rstd, eps, x=None, res=None, y=None, p_pre=0.0, p_post=0.0,
             site_pre=0, site_post=0, rng=None, ids=None, segs=None, T=0, type_vocab=2, word=None, pos=None, tt=None,
             tt_ext=None, loc=None, w_loc=None, b_loc=None, pos_offset=0):
    d = L.LnDesc()
    d.mode, d.dtype, d.M, d.H = mode, dtype, M, H
    d.x, d.ldx = _p(x), (x.stride(-2) if x is not None else 0)
    d.res, d.ldres = _p(res), (res.stride(-2) if res is not None else 0)
    d.gamma, d.beta, d.eps = _p(gamma), _p(beta), eps
    d.y, d.ldy = _p(y), (y.stride(-2) if y is not None else 0)
    d.mean, d.rstd = _p(mean), _p(rstd)
    d.p_pre, d.p_post, d.site_pre, d.site_post = p_pre, p_post, site_pre, site_post
    d.rng = rng.ptr() if (rng is not None and (p_pre > 0 or p_post > 0)) else None
    d.ids, d.segs, d.T, d.type_vocab = _p(ids), _p(segs), T, type_vocab
    d.word, d.pos, d.tt, d.tt_ext = _p(word), _p(pos), _p(tt), _p(tt_ext)
    d.loc, d.w_loc, d.b_loc = _p(loc), _p(w_loc), _p(b_loc)
    d.pos_offset = pos_offset
    return d


def ln_fwd(**kw):
    lib = L.load()
    d = _ln_desc(**kw)
    e0 = _prof_begin()
    L.check("gstvd_ln_fwd", lib.gstvd_ln_fwd(C.byref(d), _stream()))
    _prof_end(e0, "ln_fwd", 0.0, 3.0 * d.M * d.H * (2 if d.dtype == BF16 else 4), (d.M, d.H, d.mode))


def ln_bwd_blocks(M, H=None, mode=None):
    """Number of column-partial slabs the backward kernel writes for this geometry (size `partial` with it and pass it on)."""
    if H is None:
        return int(L.load().gstvd_ln_bwd_blocks(M))
    return int(L.load().gstvd_ln_bwd_blocks_for(M, H, mode))


def ln_bwd(fwd_kw, dy, partial, dres=None, dx=None, dword=None, dpos=None, dtt=None, dtt_ext=None, nblk=0):
    lib = L.load()
    b = L.LnBwdDesc()
    b.f = _ln_desc(**fwd_kw)
    b.nblk = nblk
    b.dy, b.lddy = _p(dy), dy.stride(-2)
    b.dres, b.lddres = _p(dres), (dres.stride(-2) if dres is not None else 0)
    b.dx, b.lddx = _p(dx), (dx.stride(-2) if dx is not None else 0)
    b.partial = _p(partial)
    b.dword, b.dpos, b.dtt, b.dtt_ext = _p(dword), _p(dpos), _p(dtt), _p(dtt_ext)
    e0 = _prof_begin()
    L.check("gstvd_ln_bwd", lib.gstvd_ln_bwd(C.byref(b), _stream()))
    _prof_end(e0, "ln_bwd", 0.0, 5.0 * b.f.M * b.f.H * (2 if b.f.dtype == BF16 else 4), (b.f.M, b.f.H, b.f.mode))


def colsum_partials(partial, nblk, nvec, H, out0, out1, out2, accumulate):
    lib = L.load()
    e0 = _prof_begin()
    L.check("gstvd_colsum_partials", lib.gstvd_colsum_partials(_p(partial), nblk, nvec, H, _p(out0), _p(out1), _p(out2),
                                                               int(accumulate), _stream()))
    _prof_end(e0, "colsum_partials", 0.0, 4.0 * nblk * nvec * H)


def colsum(x, M, N, out, scratch, accumulate):
    lib = L.load()
    e0 = _prof_begin()
    L.check("gstvd_colsum", lib.gstvd_colsum(_p(x), x.stride(-2), M, N, dt(x), _p(out), _p(scratch), scratch.numel(),
                                             int(accumulate), _stream()))
    _prof_end(e0, "colsum", 0.0, float(M) * N * x.element_size())


def colsum_slabs(x, M, N, scratch):
    lib = L.load()
    e0 = _prof_begin()
    L.check("gstvd_colsum_slabs", lib.gstvd_colsum_slabs(_p(x), x.stride(-2), M, N, dt(x), _p(scratch), scratch.numel(), _stream()))
    _prof_end(e0, "colsum_slabs", 0.0, float(M) * N * x.element_size())


class ColsumBatch(object):
    """Collects column reductions (LayerNorm-backward partials, bias-gradient slabs) and runs them in ONE launch.
    The device table is rebuilt only when the (static, arena-addressed) entry list changes."""

    def __init__(self, device):
        self.device = device
        self.entries = []            # (partial_ptr, (out0,out1,out2 ptrs), nblk, stride, H, nvec, (acc0,acc1,acc2))
        self.targets, self.keep = set(), []
        self.slabs, self.slab_dtype, self.slab_cache = [], None, {}
        self.cache = {}

    def add_slabs(self, x, M, N, scratch, out, accumulate):
        """out[c] (+)= sum_m x[m, c]: queue the 64-row slab stage and the final reduction (both batched at flush)."""
        nslab = (M + 63) // 64
        self.slabs.append((x.data_ptr(), scratch.data_ptr(), x.stride(-2), M, N))
        self.slab_dtype = dt(x)
        self.keep.append((x, scratch))
        self.add(scratch, (out, None, None), nslab, N, N, 1, (accumulate, False, False))

    def add(self, partial, outs, nblk, stride, H, nvec, accs):
        ptrs = [o.data_ptr() for o in outs if o is not None]
        if any(p in self.targets for p in ptrs):      # same output twice (shared embedding LayerNorm): keep launch order
            self.flush()
        self.keep.append((partial, outs))
        self.targets.update(ptrs)
        self.entries.append((partial.data_ptr(), tuple(o.data_ptr() if o is not None else 0 for o in outs), nblk, stride, H,
                             nvec, tuple(int(bool(a)) for a in accs)))

    def reset(self):
        self.entries, self.targets, self.keep, self.slabs = [], set(), [], []

    def _flush_slabs(self):
        key = tuple(self.slabs)
        hit = self.slab_cache.get(key)
        if hit is None:
            arr = (L.SlabEntry * len(key))()
            blk = 0
            for e, (xp, sp, ldx, M, N) in zip(arr, key):
                e.x, e.scratch, e.ldx, e.M, e.N, e.blk0 = xp, sp, ldx, M, N, blk
                blk += ((M + 63) // 64) * ((N + 255) // 256)
            hit = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(key), blk)
            if len(self.slab_cache) > 64:
                self.slab_cache.clear()
            self.slab_cache[key] = hit
        tab, n, blocks = hit
        lib = L.load()
        e0 = _prof_begin()
        L.check("gstvd_colsum_slabs_batched", lib.gstvd_colsum_slabs_batched(tab.data_ptr(), n, blocks, self.slab_dtype, _stream()))
        _prof_end(e0, "colsum_slabs_batched", 0.0, 0.0)
        self.slabs = []

    def flush(self):
        if not self.entries:
            return
        if self.slabs:
            self._flush_slabs()
        key = tuple(self.entries)
        hit = self.cache.get(key)
        if hit is None:
            arr = (L.ColsumEntry * len(key))()
            blk = 0
            for e, (pp, outs, nblk, stride, H, nvec, accs) in zip(arr, key):
                e.partial = pp
                for j in range(3):
                    e.out[j] = outs[j] or None
                    e.accumulate[j] = accs[j]
                e.nblk, e.stride, e.H, e.nvec, e.blk0 = nblk, stride, H, nvec, blk
                blk += (nvec * H + 63) // 64
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            hit = (host.to(self.device), len(key), blk)
            if len(self.cache) > 64:
                self.cache.clear()
            self.cache[key] = hit
        tab, n, blocks = hit
        lib = L.load()
        e0 = _prof_begin()
        L.check("gstvd_colsum_batched", lib.gstvd_colsum_batched(tab.data_ptr(), n, blocks, _stream()))
        _prof_end(e0, "colsum_batched", 0.0, 0.0)
        self.entries, self.targets, self.keep = [], set(), []


def locgrad(dh, loc, M, H, dw_loc, accumulate):
    lib = L.load()
    L.check("gstvd_locgrad", lib.gstvd_locgrad(_p(dh), dh.stride(-2), _p(loc), M, H, dt(dh), _p(dw_loc), int(accumulate),
                                               _stream()))


KEEP_BITS = 1     # 0 (tests / tools patch the attribute): the backward hashes its dropout draws again (profiles/r05_attn_keep_bits_ab.txt)


def attn_keep_bits_shape(B, nh, Lq, Lk, d, dtype, causal, drop_p):
    """int64 element count of the keep-bit buffer that attn_desc(drop_bits=...) takes, or 0 when the shape's backward does not read
    one (the one-pass kernel's range: bf16, d = 64, no causal mask, 64 < keys <= 256, 64 <= queries <= 1024, dropout on)."""
    if not (KEEP_BITS and dtype == torch.bfloat16 and d == 64 and not causal and drop_p > 0 and 64 < Lk <= 256 and 64 <= Lq <= 1024):
        return 0
    return B * nh * ((Lq + 15) // 16) * ((Lk + 15) // 16) * 4


def attn_desc(Q, K, V, O, LSE, key_mask, B, nh, Lq, Lk, d, *, causal=False, mask_neg=-10000.0, scale=None, drop_p=0.0,
              site=0, rng=None, ldq=None, ldk=None, ldv=None, ldo=None, kv_group=1, q_bstride=0, kv_bstride=0, drop_bits=None):
    a = L.AttnDesc()
    a.Q, a.K, a.V, a.O, a.LSE, a.key_mask = _p(Q), _p(K), _p(V), _p(O), _p(LSE), _p(key_mask)
    a.ldq = Q.stride(-2) if ldq is None else ldq
    a.ldk = K.stride(-2) if ldk is None else ldk
    a.ldv = V.stride(-2) if ldv is None else ldv
    a.ldo = O.stride(-2) if ldo is None else ldo
    a.B, a.nh, a.Lq, a.Lk, a.d, a.causal, a.dtype = B, nh, Lq, Lk, d, int(causal), dt(Q)
    a.mask_neg = mask_neg
    a.scale = (1.0 / (d ** 0.5)) if scale is None else scale
    a.dropout_p, a.site = drop_p, site
    a.rng = rng.ptr() if (rng is not None and drop_p > 0) else None
    a.kv_group, a.q_bstride, a.kv_bstride = kv_group, q_bstride, kv_bstride
    if drop_bits is not None:
        need = B * nh * ((Lq + 15) // 16) * ((Lk + 15) // 16) * 4
        if drop_bits.dtype != torch.int64 or drop_bits.numel() < need or not drop_bits.is_contiguous():
            raise L.GstvdError("attn_desc: drop_bits must be a contiguous int64 tensor of >= %d elements" % need)
        a.drop_bits = _p(drop_bits)
    return a


def attn_fwd(a):
    lib = L.load()
    e0 = _prof_begin()
    L.check("gstvd_attn_fwd", lib.gstvd_attn_fwd(C.byref(a), _stream()))
    _prof_end(e0, "attn_fwd_d%d" % a.d, 4.0 * a.B * a.nh * a.Lq * a.Lk * a.d, 0.0, (a.B, a.nh, a.Lq, a.Lk))


def attn_bwd(a, dO, dQ, dK, dV, delta, lddo=None, lddq=None, lddk=None, lddv=None):
    lib = L.load()
    a.dO, a.dQ, a.dK, a.dV, a.delta = _p(dO), _p(dQ), _p(dK), _p(dV), _p(delta)
    a.lddo = dO.stride(-2) if lddo is None else lddo
    a.lddq = dQ.stride(-2) if lddq is None else lddq
    a.lddk = dK.stride(-2) if lddk is None else lddk
    a.lddv = dV.stride(-2) if lddv is None else lddv
    e0 = _prof_begin()
    L.check("gstvd_attn_bwd", lib.gstvd_attn_bwd(C.byref(a), _stream()))
    _prof_end(e0, "attn_bwd_d%d" % a.d, 14.0 * a.B * a.nh * a.Lq * a.Lk * a.d, 0.0, (a.B, a.nh, a.Lq, a.Lk))


def ce_fwd(logits, labels, M, V, row_loss, lse, stats, ignore_index=0):
    lib = L.load()
    e0 = _prof_begin()
    L.check("gstvd_ce_fwd", lib.gstvd_ce_fwd(_p(logits), logits.stride(-2), _p(labels), M, V, ignore_index, dt(logits),
                                             _p(row_loss), _p(lse), _p(stats), _stream()))
    _prof_end(e0, "ce_fwd", 0.0, float(M) * V * logits.element_size(), (M, V))


def ce_bwd(logits, labels, lse, stats, gscale, mean, M, V, dlogits, ignore_index=0):
    lib = L.load()
    e0 = _prof_begin()
    L.check("gstvd_ce_bwd", lib.gstvd_ce_bwd(_p(logits), logits.stride(-2), _p(labels), _p(lse), _p(stats), _p(gscale),
                                             int(mean), M, V, ignore_index, dt(logits), _p(dlogits), dlogits.stride(-2),
                                             _stream()))
    _prof_end(e0, "ce_bwd", 0.0, float(M) * V * (logits.element_size() + dlogits.element_size()), (M, V))


SAMPLE_MAX_TOP_K = 1 << 30  # (ABI 6: no limit any more -- k <= 16 walks the distinct values from the top, larger k bisects; top-p is in the kernel too)
SAMPLE_MAX_VOCAB = 31 * 1024   # a row of scaled logits lives in one CU's LDS (and 31 registers per thread)


SPECIAL_TOKEN_IDS = (0, 100, 101, 102, 103)      # utils/decoding_utils.py:38 (the default no reference caller overrides)


def sample_topk(logits, temperature, top_k, u, out, banned=None, ngram=None, top_p=0.0):
    """One sampling step (gstvd_sample_topk): out[b] <- inverse-CDF draw from softmax(top_p(top_k(logits / temperature, banned -> -inf))).
    top_p in (0, 1): nucleus filtering (utils/decoding_utils.py:22-34) inside the launch; 0 / >= 1: off.
    logits [B, V] fp32 / bf16 (row stride free); u [B] fp32 in (0, 1); out: int64 view with B elements (any stride, e.g. a
    column of the id buffer); banned: None or bool / uint8 [B, >= V].
    ngram = (hist [B, T] int64, ids_tm [L, B] int64 time-major, cur_len, n[, special ids]): the n-gram filter of
    utils/decoding_utils.py:38-77 inside the same launch (the row's last n-1 ids are ids_tm[cur_len-(n-1) : cur_len, b])."""
    lib = L.load()
    Bn, V = logits.shape
    d = L.SampleDesc()
    d.logits, d.ld, d.dtype, d.B, d.V, d.top_k, d.temperature = _p(logits), logits.stride(0), dt(logits), Bn, V, int(top_k), float(temperature)
    d.top_p = float(top_p)
    if u.dtype != torch.float32 or not u.is_contiguous() or u.numel() != Bn:
        raise L.GstvdError("sample_topk: u must be a contiguous fp32 vector of B uniforms")
    if out.dtype != torch.int64 or out.numel() != Bn or logits.stride(1) != 1:
        raise L.GstvdError("sample_topk: out must hold B int64 ids; logits rows must be dense")
    d.u, d.out, d.out_stride = _p(u), _p(out), (out.stride(0) if out.dim() else 1)
    if banned is not None:
        if banned.dtype not in (torch.bool, torch.uint8) or banned.stride(1) != 1 or banned.shape[1] < V:
            raise L.GstvdError("sample_topk: banned must be bool / uint8 [B, >= V] with dense rows")
        d.banned, d.banned_ld = _p(banned), banned.stride(0)
    if ngram is not None and int(ngram[3]) > 0:
        hist, ids_tm, cur_len, n = ngram[:4]
        special = tuple(ngram[4]) if len(ngram) > 4 else SPECIAL_TOKEN_IDS
        if hist.dtype != torch.int64 or ids_tm.dtype != torch.int64 or hist.stride(1) != 1 or ids_tm.stride(1) != 1 or hist.shape[0] != Bn:
            raise L.GstvdError("sample_topk: hist [B, T] and ids_tm [L, B] must be int64 with dense rows")
        if len(special) > 8 or ids_tm.shape[0] < cur_len:
            raise L.GstvdError("sample_topk: at most 8 special ids; ids_tm must hold cur_len positions")
        d.hist, d.hist_ld, d.hist_T, d.ngram = _p(hist), hist.stride(0), hist.shape[1], int(n)
        d.ids_tm, d.ids_stride, d.cur_len, d.n_special = _p(ids_tm), ids_tm.stride(0), int(cur_len), len(special)
        for i, t in enumerate(special):
            d.special[i] = int(t)
    L.check("gstvd_sample_topk", lib.gstvd_sample_topk(C.byref(d), _stream()))


def answer_scores(logits, lse, dec_ids, rows, U, scores):
    lib = L.load()
    L.check("gstvd_answer_scores", lib.gstvd_answer_scores(_p(logits), logits.stride(-2), _p(lse), _p(dec_ids), rows, U,
                                                           dt(logits), _p(scores), _stream()))


def vl_split(d_enc, B, R, T, H, d_v, d_t, p, site_v, site_t, rng):
    lib = L.load()
    L.check("gstvd_vl_split", lib.gstvd_vl_split(_p(d_enc), B, R, T, H, dt(d_enc), _p(d_v), _p(d_t), p, site_v, site_t,
                                                 rng.ptr() if (rng is not None and p > 0) else None, _stream()))


def cast(src, dst, n=None):
    lib = L.load()
    n = src.numel() if n is None else n
    L.check("gstvd_cast", lib.gstvd_cast(_p(src), dt(src), _p(dst), dt(dst), n, _stream()))
    return dst


class CastRanges(object):
    """A fixed list of ranges [(start, length)] of two flat buffers with the same indexing, uploaded once (gstvd_cast_ranges)."""

    def __init__(self, ranges, device):
        self.ranges = [(int(a), int(n)) for a, n in ranges if n > 0]
        if any(a % 4 for a, _ in self.ranges):
            raise L.GstvdError("cast_ranges: range starts must be multiples of 4 elements")
        blk0, b = [], 0
        for _, n in self.ranges:
            blk0.append(b)
            b += (n + 1023) // 1024
        blk0.append(b)
        self.blocks, self.elems = b, sum(n for _, n in self.ranges)
        flat = [v for r in self.ranges for v in r]
        self.tab = torch.tensor(flat, dtype=torch.int64).to(device) if flat else None
        self.blk0 = torch.tensor(blk0, dtype=torch.int32).to(device)

    def run(self, src, dst):
        """dst[e] = bf16(src[e]) over the ranges; src fp32, dst bf16, both indexed alike."""
        if not self.ranges:
            return
        lib = L.load()
        L.check("gstvd_cast_ranges", lib.gstvd_cast_ranges(_p(src), _p(dst), self.tab.data_ptr(), self.blk0.data_ptr(), len(self.ranges),
                                                           self.blocks, _stream()))


def scale_(x, factor):
    lib = L.load()
    L.check("gstvd_scale", lib.gstvd_scale(_p(x), _p(factor), x.numel(), _stream()))


def dropout_mask(n, p, site, rng, device):
    lib = L.load()
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.check("gstvd_dropout_mask", lib.gstvd_dropout_mask(_p(out), n, p, site, rng.ptr(), _stream()))
    return out


def adamw(param, grad, m, v, shadow, seg_end, hp, step, beta1=0.9, beta2=0.999, eps=1e-6, grad_scale=1.0, begin=0, end=None,
          grad_origin=0):
    """Fused AdamW over flat elements [begin, end) of the buffers.  `grad` is the flat fp32 gradient buffer, or a bf16
    tensor holding flat elements [grad_origin, grad_origin + grad.numel()) (the all-reduced compressed slice)."""
    lib = L.load()
    n = param.numel() if end is None else end
    e0 = _prof_begin()
    if grad.dtype == torch.bfloat16:
        if grad_origin + grad.numel() < n:
            raise ValueError("bf16 gradient slice does not cover [begin, end)")
        L.check("gstvd_adamw_bf16grad", lib.gstvd_adamw_bf16grad(_p(param), _p(grad), grad_origin, _p(m), _p(v), _p(shadow), n,
                                                                 _p(seg_end), _p(hp), seg_end.numel(), beta1, beta2, eps, _p(step),
                                                                 grad_scale, begin, _stream()))
    else:
        L.check("gstvd_adamw", lib.gstvd_adamw(_p(param), _p(grad), _p(m), _p(v), _p(shadow), n, _p(seg_end), _p(hp),
                                               seg_end.numel(), beta1, beta2, eps, _p(step), grad_scale, begin, _stream()))
    _prof_end(e0, "adamw", 0.0, (n - begin) * (30.0 if shadow is not None else 28.0))


def adamw_blocks(param, grad, m, v, shadow, seg_end, hp, step, blocks, seg_skip, beta1=0.9, beta2=0.999, eps=1e-6, grad_scale=1.0,
                 begin=0, end=None):
    """AdamW on the listed 1024-element blocks (int32 device tensor of absolute block indices), clipped to [begin, end), leaving
    the segments flagged in `seg_skip` (uint8 device tensor, one per segment) alone: the remainder behind a weight-gradient
    launch that updated its weights itself (GemmGroup.flush(fuse=...))."""
    lib = L.load()
    n = param.numel() if end is None else end
    e0 = _prof_begin()
    L.check("gstvd_adamw_blocks", lib.gstvd_adamw_blocks(_p(param), _p(grad), _p(m), _p(v), _p(shadow), n, _p(seg_end), _p(hp),
                                                         seg_end.numel(), beta1, beta2, eps, _p(step), grad_scale, begin, _p(blocks),
                                                         blocks.numel(), _p(seg_skip), _stream()))
    _prof_end(e0, "adamw", 0.0, blocks.numel() * 1024 * (30.0 if shadow is not None else 28.0))


def gemm_ln_rows():
    """Rows per block of the LayerNorm-folded GEMMs (their backward writes one [3][H] partial slab per block)."""
    return int(L.load().gstvd_gemm_ln_rows_per_block())


def _gemm_desc_noA(B, C_out, M, N, K, b_km, bias, addend, aux, epi):
    d = L.GemmDesc()
    d.A, d.B, d.C = None, _p(B), _p(C_out)
    d.bias, d.addend, d.aux = _p(bias), _p(addend), _p(aux)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = 0, B.stride(-2), C_out.stride(-2)
    d.ldadd = addend.stride(-2) if addend is not None else 0
    d.ldaux = aux.stride(-2) if aux is not None else 0
    d.batch, d.dtype_in, d.dtype_out, d.alpha = 1, dt(B), dt(C_out), 1.0
    d.a_kmajor, d.b_kmajor = 0, int(b_km)
    if bias is not None:
        epi |= EPI_BIAS
    if addend is not None:
        epi |= EPI_ADD
    d.epilogue = epi
    return d


def gemm_ln_fwd(ln_kw, B, C_out, N, *, b_km=False, bias=None, addend=None, aux=None, epi=0):
    """C[M, N] = epi(LN(ln_kw) . B^T) in ONE launch (gstvd_gemm_ln_fwd); also writes ln_kw['y'] / mean / rstd as ln_fwd would.
    Raises GstvdError(GSTVD_E_UNSUPPORTED) for shapes outside the kernel's range -- callers check `gemm_ln_ok` first."""
    lib = L.load()
    ld = _ln_desc(**ln_kw)
    d = _gemm_desc_noA(B, C_out, ld.M, N, ld.H, b_km, bias, addend, aux, epi)
    e0 = _prof_begin()
    L.check("gstvd_gemm_ln_fwd", lib.gstvd_gemm_ln_fwd(C.byref(d), C.byref(ld), _stream()))
    _prof_end(e0, "gemm_ln_fwd", 2.0 * ld.M * N * ld.H, float(3 * ld.M * ld.H * 2 + N * ld.H * 2 + ld.M * N * C_out.element_size()),
              (ld.M, N, ld.H, 1))
    return C_out


def gemm_ln_bwd(ln_kw, dy, partial, nblk, B, C_out, N, *, dres=None, dx=None, b_km=True, addend=None, aux=None, epi=0):
    """C[M, N] = epi(dx . B) with dx = the LayerNorm backward of ln_kw under dy, in ONE launch (gstvd_gemm_ln_bwd); also writes
    dres, dx and the [nblk, 3, H] column partials (nblk = ceil(M / gemm_ln_rows()))."""
    lib = L.load()
    b = L.LnBwdDesc()
    b.f = _ln_desc(**ln_kw)
    b.nblk = nblk
    b.dy, b.lddy = _p(dy), dy.stride(-2)
    b.dres, b.lddres = _p(dres), (dres.stride(-2) if dres is not None else 0)
    b.dx, b.lddx = _p(dx), (dx.stride(-2) if dx is not None else 0)
    b.partial = _p(partial)
    d = _gemm_desc_noA(B, C_out, b.f.M, N, b.f.H, b_km, None, addend, aux, epi)
    e0 = _prof_begin()
    L.check("gstvd_gemm_ln_bwd", lib.gstvd_gemm_ln_bwd(C.byref(d), C.byref(b), _stream()))
    _prof_end(e0, "gemm_ln_bwd", 2.0 * b.f.M * N * b.f.H, float(5 * b.f.M * b.f.H * 2 + N * b.f.H * 2 + b.f.M * N * C_out.element_size()),
              (b.f.M, N, b.f.H, 1))
    return C_out


LN_FOLD_MAX_H = 768


def gemm_ln_ok(M, N, H, dtype):
    """Shapes the engine gives to the LayerNorm-folded GEMMs (csrc/gemm_rows.hip): bf16, H = K a multiple of 64 up to
    ops.LN_FOLD_MAX_H = 768 (the decoder's sites), up to 640 rows (beyond that the plain kernels' larger tiles win), at least
    five 128-column tiles.  The kernel itself takes H up to 1024 since round 5 (the vision stream's width; LN_FOLD_MAX_H = 1024
    sends its 36 sites there): bit-identical, but an 80 KB / 512-thread workgroup does not fit beside the text stream's
    workgroups the way the 48 KB 64-tile GEMM + the LayerNorm kernel do -- 12.35 -> 12.66 ms per step
    (profiles/r05_ln_fold_vision_ab.txt), so the default keeps the vision sites on the separate kernels."""
    if not (dtype == torch.bfloat16 and H % 64 == 0 and H <= LN_FOLD_MAX_H and M <= 640 and N >= 640 and N % 8 == 0):
        return False
    # one round of the chip: two workgroups (16 rows x 128 columns) fit a CU; beyond 512 the second round doubles the launch
    # (N = 3072 at 400 rows: 22 us forward / 38 us backward against 23 / 26 us for the two separate kernels)
    return ((M + 15) // 16) * ((N + 127) // 128) <= 512
