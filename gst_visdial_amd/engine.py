"""Execution engine of the enc_dec_a hot path on one MI355X.

Replaces, for the modules of `modules.py`, the eager PyTorch arithmetic of
    models/vilbert_dialog.py:324-912,1325-1427   (embeddings, two-stream encoder schedule)
    models/visual_dialog_model.py:24-72,123-135 (glue, VLFusion)
    models/visual_dialog_decoder.py:33-86,219-339 + transformers-4.16.2 BertEncoder (decoder, LM head, CE)
and autograd's backward of all of it (train_gen.py:324) by a fixed schedule of HIP kernels called through
the C ABI (`ops`).  Design points:

  * parameters live in ONE flat fp32 buffer `P` (forward order), gradients in one flat fp32 buffer `G`,
    bf16 shadow weights (throughput mode) in one flat bf16 buffer `S`; the nn.Parameters are views.
    Q/K/V (and the co-attention / cross-attention K,V of all decoder layers) are laid out contiguously
    so each projection group is ONE GEMM, and a data-parallel all-reduce is a handful of large slices;
  * activations come from a bump arena that is rewound every step -> static addresses (hipGraph friendly),
    no allocator traffic; nothing of size [Lq, Lk] is ever stored (attention saves only LSE);
  * forward records a tape of backward closures; backward replays it in reverse.  Residual gradients are
    accumulated in the dgrad GEMM epilogue, GELU' in the dgrad epilogue, bias gradients come out of the
    LayerNorm backward partial sums; dropout masks are regenerated from (seed, step, site, index);
  * the 42 parameters that get no gradient in enc_dec mode (poolers, cls heads, q_dense*, sep_embeddings;
    models/vilbert_dialog.py:1400-1401,1482-1487) are never touched: their forward is dead compute in the
    reference and skipping it changes no output.

Numerics: precision 'fp32' runs every GEMM on the exact-fp32 MFMA (parity gate: logits within 1e-4 of the
oracle); 'bf16' stores activations/weights in bf16 with fp32 accumulation and fp32 LN/softmax/CE statistics.
"""
import os
import weakref

import torch

from . import ops
from . import _lib as _libmod
from .config import encoder_schedule
from ._lib import GstvdError, EPI_GELU, EPI_DGELU, LN_RESID, LN_EMBED, LN_IMAGE


EARLY_WGRAD = 1     # 0 (tests / tools patch the attribute): the final grouped weight-gradient launch waits for the embedding backward


def _round_up(x, m):
    return (x + m - 1) // m * m


class Act(object):
    """An activation [M, N] in the arena plus (during backward) its gradient."""
    __slots__ = ("t", "g", "M", "N", "gelu_aux", "bias_done", "prod", "dgrad_done")

    def __init__(self, t, M, N):
        self.t, self.g, self.M, self.N = t, None, M, N
        self.gelu_aux, self.bias_done = None, False
        self.prod, self.dgrad_done = None, False      # the Linear that produced it (input, weight, K_in, need_dx); see _ln_bwd


class Arena(object):
    """Bump allocator over large device chunks; `reset()` rewinds, so a fixed call sequence gets fixed addresses
    (hipGraph friendly).  The request sequence of a step is identical from step to step, so the tensor views are
    memoised by sequence index: steady-state allocation is a list lookup, no tensor construction."""

    _ESZ = {torch.float32: 4, torch.bfloat16: 2, torch.int64: 8, torch.uint8: 1, torch.int32: 4}

    def __init__(self, device, chunk_bytes=1 << 28):
        self.device, self.chunk_bytes = device, chunk_bytes
        self.chunks, self.ci, self.off = [], 0, 0
        self.memo, self.seq = [], 0

    def reset(self):
        self.ci, self.off, self.seq = 0, 0, 0

    def rewind(self, mark):
        """Back to a position returned by `mark()` (decode loops reuse the same scratch every step)."""
        self.ci, self.off, self.seq = mark

    def mark(self):
        return (self.ci, self.off, self.seq)

    def alloc(self, numel, dtype, shape=None):
        nbytes = _round_up(numel * self._ESZ[dtype], 256)
        i = self.seq
        self.seq = i + 1
        if i < len(self.memo):
            m = self.memo[i]
            if m[0] == numel and m[1] is dtype and m[2] == shape and m[3] == self.ci and m[4] == self.off:
                self.ci, self.off = m[5], m[6]
                return m[7]
            del self.memo[i:]                 # the sequence diverged (different shapes): rebuild from here
        ci0, off0 = self.ci, self.off
        while True:
            if self.ci >= len(self.chunks):
                self.chunks.append(torch.empty(max(self.chunk_bytes, nbytes), dtype=torch.uint8, device=self.device))
            c = self.chunks[self.ci]
            if self.off + nbytes <= c.numel():
                out = c[self.off:self.off + nbytes].view(dtype)[:numel]
                if shape is not None:
                    out = out.view(shape)
                self.off += nbytes
                self.memo.append((numel, dtype, shape, ci0, off0, self.ci, self.off, out))
                return out
            self.ci, self.off = self.ci + 1, 0


class FlatParams(object):
    """Flat storage plan.  `slots[name] = (offset, shape)` are engine views (possibly fused groups of several
    nn.Parameters); every live nn.Parameter becomes a view of `P` and its `.grad` a view of `G`."""

    def __init__(self, model, precision):
        enc_cfg, dec_cfg = model.encoder.config, model.decoder.config
        bert = model.encoder.bert_pretrained.bert
        gen = model.decoder.decoder
        self.slots, self.items, self.pads = {}, [], []
        self.placed = {}
        self.off = 0
        H, Hv, Hb = enc_cfg.hidden_size, enc_cfg.v_hidden_size, enc_cfg.bi_hidden_size
        V = dec_cfg.vocab_size
        self.Vp = _round_up(V, 64)
        lm_w = gen.lm_head.decoder.weight

        def place(name, params, shape=None, pad_rows_to=None):
            """Lay `params` out back to back under one fused slot `name`."""
            self.off = _round_up(self.off, 64)
            start = self.off
            for p in params:
                if id(p) in self.placed:
                    raise GstvdError("parameter shared between two fused groups: " + name)
                self.placed[id(p)] = self.off
                self.items.append((p, self.off))
                self.off += p.numel()
            if pad_rows_to is not None:
                cols = params[0].shape[1] if params[0].dim() == 2 else 1
                want = pad_rows_to * cols
                self.pads.append((self.off, start + want))
                self.off = start + want
            n = self.off - start
            if shape is None:
                shape = tuple(params[0].shape) if len(params) == 1 and pad_rows_to is None else (n,)
            self.slots[name] = (start, shape)

        def emb(prefix, mod):
            w = mod.word_embeddings.weight
            place(prefix + ".word", [w], shape=(self.Vp if w is lm_w else w.shape[0], w.shape[1]),
                  pad_rows_to=self.Vp if w is lm_w else None)
            place(prefix + ".pos", [mod.position_embeddings.weight])
            place(prefix + ".tt", [mod.token_type_embeddings.weight])
            place(prefix + ".tte", [mod.token_type_embeddings_extension.weight])
            place(prefix + ".ln.w", [mod.LayerNorm.weight])
            place(prefix + ".ln.b", [mod.LayerNorm.bias])

        def attn_out_ffn(p, lay, hid, inter):
            place(p + ".ao.w", [lay.attention.output.dense.weight]); place(p + ".ao.b", [lay.attention.output.dense.bias])
            place(p + ".ln1.w", [lay.attention.output.LayerNorm.weight]); place(p + ".ln1.b", [lay.attention.output.LayerNorm.bias])

        def ffn(p, inter_mod, out_mod, tag_i, tag_o, tag_ln):
            place(p + tag_i + ".w", [inter_mod.dense.weight]); place(p + tag_i + ".b", [inter_mod.dense.bias])
            place(p + tag_o + ".w", [out_mod.dense.weight]); place(p + tag_o + ".b", [out_mod.dense.bias])
            place(p + tag_ln + ".w", [out_mod.LayerNorm.weight]); place(p + tag_ln + ".b", [out_mod.LayerNorm.bias])

        def qkv(p, tag, q, k, v, hid_out, hid_in):
            place(p + tag + ".w", [q.weight, k.weight, v.weight], shape=(3 * hid_out, hid_in))
            place(p + tag + ".b", [q.bias, k.bias, v.bias], shape=(3 * hid_out,))

        def bert_layer(p, lay, hid, inter):
            s = lay.attention.self
            qkv(p, ".qkv", s.query, s.key, s.value, hid, hid)
            attn_out_ffn(p, lay, hid, inter)
            ffn(p, lay.intermediate, lay.output, ".fi", ".fo", ".ln2")

        self.enc_emb = bert.embeddings
        self.dec_emb = gen.bert.embeddings
        emb("emb", self.enc_emb)
        ve = bert.v_embeddings
        place("vemb.img.w", [ve.image_embeddings.weight]); place("vemb.img.b", [ve.image_embeddings.bias])
        place("vemb.loc.w", [ve.image_location_embeddings.weight]); place("vemb.loc.b", [ve.image_location_embeddings.bias])
        place("vemb.ln.w", [ve.LayerNorm.weight]); place("vemb.ln.b", [ve.LayerNorm.bias])
        self.marks = {}
        for kind, i in encoder_schedule(enc_cfg):
            self.marks[(kind, i)] = _round_up(self.off, 64)
            if kind == "t":
                bert_layer("t%d" % i, bert.encoder.layer[i], H, enc_cfg.intermediate_size)
            elif kind == "v":
                bert_layer("v%d" % i, bert.encoder.v_layer[i], Hv, enc_cfg.v_intermediate_size)
            else:
                c, p = bert.encoder.c_layer[i], "c%d" % i
                b = c.biattention
                qkv(p, ".qkv1", b.query1, b.key1, b.value1, Hb, Hv)
                qkv(p, ".qkv2", b.query2, b.key2, b.value2, Hb, H)
                o = c.biOutput
                place(p + ".d1.w", [o.dense1.weight]); place(p + ".d1.b", [o.dense1.bias])
                place(p + ".ln1.w", [o.LayerNorm1.weight]); place(p + ".ln1.b", [o.LayerNorm1.bias])
                place(p + ".d2.w", [o.dense2.weight]); place(p + ".d2.b", [o.dense2.bias])
                place(p + ".ln2.w", [o.LayerNorm2.weight]); place(p + ".ln2.b", [o.LayerNorm2.bias])
                ffn(p, c.v_intermediate, c.v_output, ".vfi", ".vfo", ".vln")
                ffn(p, c.t_intermediate, c.t_output, ".tfi", ".tfo", ".tln")
        self.marks["vlf"] = _round_up(self.off, 64)
        place("vlf.v.w", [model.vlfusion.fc_v.weight]); place("vlf.v.b", [model.vlfusion.fc_v.bias])
        place("vlf.l.w", [model.vlfusion.fc_l.weight]); place("vlf.l.b", [model.vlfusion.fc_l.bias])
        self.marks["dec"] = _round_up(self.off, 64)
        if self.dec_emb is not self.enc_emb:
            emb("demb", self.dec_emb)
        layers = gen.bert.encoder.layer
        Hd, L = dec_cfg.hidden_size, len(layers)
        kvw, kvb = [], []
        for lay in layers:
            cs = lay.crossattention.self
            kvw += [cs.key.weight, cs.value.weight]
            kvb += [cs.key.bias, cs.value.bias]
        place("dec.ckv.w", kvw, shape=(2 * L * Hd, Hd))
        place("dec.ckv.b", kvb, shape=(2 * L * Hd,))
        for i, lay in enumerate(layers):
            p = "d%d" % i
            self.marks[("d", i)] = _round_up(self.off, 64)
            s = lay.attention.self
            qkv(p, ".qkv", s.query, s.key, s.value, Hd, Hd)
            attn_out_ffn(p, lay, Hd, dec_cfg.intermediate_size)
            c = lay.crossattention
            place(p + ".cq.w", [c.self.query.weight]); place(p + ".cq.b", [c.self.query.bias])
            place(p + ".co.w", [c.output.dense.weight]); place(p + ".co.b", [c.output.dense.bias])
            place(p + ".ln2.w", [c.output.LayerNorm.weight]); place(p + ".ln2.b", [c.output.LayerNorm.bias])
            ffn(p, lay.intermediate, lay.output, ".fi", ".fo", ".ln3")
        self.marks["lm"] = _round_up(self.off, 64)
        if id(lm_w) in self.placed:
            wname = "emb.word" if lm_w is self.enc_emb.word_embeddings.weight else "demb.word"
            self.slots["lm.w"] = self.slots[wname]
        else:
            place("lm.w", [lm_w], shape=(self.Vp, lm_w.shape[1]), pad_rows_to=self.Vp)
        place("lm.b", [gen.lm_head.bias], shape=(self.Vp,), pad_rows_to=self.Vp)
        self.n_live = _round_up(self.off, 64)
        self.live = [p for p, _ in self.items]
        live_ids = set(id(p) for p in self.live)
        self.dead = [p for p in model.parameters() if id(p) not in live_ids]
        self.precision = precision
        self.P = self.G = self.S = self.D = None
        self.stale_guard = None       # callable -> True while fp32 masters of other ranks' shards are old (set by the engine)

    # -- materialise on the device the parameters currently live on ------------------------------------
    def materialize(self, device):
        P = torch.zeros(self.n_live, dtype=torch.float32, device=device)
        for p, off in self.items:
            P[off:off + p.numel()].copy_(p.data.reshape(-1))
        nd = sum(p.numel() for p in self.dead)
        D = torch.empty(max(nd, 1), dtype=torch.float32, device=device)
        o = 0
        for p in self.dead:
            D[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = D[o:o + p.numel()].view(p.shape)
            o += p.numel()
        for p, off in self.items:
            p.data = P[off:off + p.numel()].view(p.shape)
        self.P, self.D = P, D
        self.G = torch.zeros(self.n_live, dtype=torch.float32, device=device)
        self.S = torch.empty(self.n_live, dtype=torch.bfloat16, device=device) if self.precision == "bf16" else None
        self.grad_views = [self.G[off:off + p.numel()].view(p.shape) for p, off in self.items]
        self.ptrs = [(p, P[off:off + p.numel()].data_ptr()) for p, off in self.items]
        self.shadow_version = None
        self.device = device

    def is_materialized(self):
        if self.P is None:
            return False
        for p, ptr in self.ptrs:
            if p.data_ptr() != ptr:
                return False
        return True

    def version(self):
        return sum(p._version for p in self.live)

    def refresh_shadow(self, force=False):
        if self.S is None:
            return
        v = self.version()
        if force or v != self.shadow_version:
            if self.shadow_version is not None and self.stale_guard is not None and self.stale_guard():
                # sharded optimizer (pipeline.BackwardPipeline(shard_update=True)): P holds current master weights only for this
                # rank's shards, S holds the freshly GATHERED shadows of all of them -- a re-cast would replace other ranks' current
                # bf16 weights by this rank's old masters and the ranks would diverge silently
                from ._lib import GstvdError
                raise GstvdError("a parameter was modified in place while the optimizer is sharded over the ranks: the bf16 shadow "
                                 "weights cannot be re-derived from this rank's fp32 masters (current only for its own shards).  "
                                 "Call pipe.sync_master() on EVERY rank before editing parameters.")
            ops.cast(self.P, self.S)
            self.shadow_version = v

    def fp32_read_ranges(self):
        """Sorted, disjoint flat ranges [x, y) whose fp32 values the forward reads directly (Engine.Pv: biases, LayerNorm gain /
        bias, the embedding tables, the image-location projection) -- everything in [0, n_live) that is not EXCLUSIVELY a GEMM
        weight (Engine.W: read from the bf16 shadow buffer in bf16 mode).  pipeline.BackwardPipeline(shard_update=True) gathers
        these in fp32; the GEMM weights only travel as bf16 shadows."""
        def numel(shape):
            n = 1
            for d in shape:
                n *= d
            return n
        non_gemm = ("ln.w", "ln1.w", "ln2.w", "ln3.w", "vln.w", "tln.w", "vemb.loc.w")
        by_start = {}
        for name, (off, shape) in self.slots.items():
            gemm = name.endswith(".w") and not name.endswith(non_gemm)
            by_start.setdefault((off, numel(shape)), []).append(gemm)
        shadow_only = sorted(k for k, flags in by_start.items() if all(flags))      # (the tied LM head aliases an embedding table: not all)
        out, pos = [], 0
        for off, n in shadow_only:
            if off > pos:
                out.append((pos, off))
            pos = max(pos, off + n)
        if pos < self.n_live:
            out.append((pos, self.n_live))
        return out

    def view(self, buf, name):
        off, shape = self.slots[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)


_DEVICE_STREAMS = {}


def device_streams(device):
    """ONE (vision, auxiliary) stream pair per device for every engine of the process.  torch hands out pool streams
    round-robin over 32 slots; a pair per Engine meant that after ~16 models (a test session, a checkpoint sweep) new
    engines aliased older engines' streams and the capture stream."""
    key = torch.device(device).index
    if key is None:                     # an index-less 'cuda': the CURRENT device, not device 0
        key = torch.cuda.current_device()
    if key not in _DEVICE_STREAMS:
        _DEVICE_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _DEVICE_STREAMS[key]


class Engine(object):
    def __init__(self, model):
        # weak: the model owns the engine, not the other way round -- no reference cycle, so dropping the model frees the
        # flat buffers, the arena and any captured decode sessions by reference counting, not at some later GC pass
        self._setup(weakref.ref(model), model.encoder.config, model.decoder.config, model.params)

    def _setup(self, model_ref, enc_cfg, dec_cfg, params):
        self._model_ref = model_ref
        self.enc_cfg, self.dec_cfg = enc_cfg, dec_cfg
        prec = params.get("amd_precision", "bf16")
        if prec not in ("bf16", "fp32"):
            raise GstvdError("params['amd_precision'] must be 'bf16' or 'fp32'")
        self.precision = prec
        self.adt = torch.bfloat16 if prec == "bf16" else torch.float32
        self.flat = None
        self.arena = None
        self.rng = None
        self._decode_sessions = {}
        self._last_decode = None
        self.anchor = None
        self._emb_span_ok = {}
        self.pipe = None               # BackwardPipeline (pipeline.py): slice-wise wgrad / all-reduce / AdamW on the aux stream
        self.tape, self.rec = [], False
        self.accumulate, self.written = False, set()
        self.stats = {}
        self._validate = True
        self.use_streams = bool(params.get("amd_streams", True))
        # LayerNorm folded into the Linear next to it for latency-bound row counts (the decoder: csrc/gemm_rows.hip)
        self.fuse_ln = (prec == "bf16" and bool(params.get("amd_fuse_ln", True)) and os.environ.get("GSTVD_FUSE_LN", "1") != "0")
        self.tag = "t"

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise GstvdError("the EncoderDecoderModel this engine belonged to is gone")
        return m

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) copies the module tree; the copy's `engine` attribute must serve the COPY, not stay bound
        (through the weak reference) to the original.  The owner is already in `memo` when its attributes are copied."""
        import copy
        orig = self._model_ref()
        owner = memo.get(id(orig))
        if owner is None:
            raise GstvdError("an Engine is copied with its EncoderDecoderModel (copy.deepcopy(model)), not on its own")
        # (the owner is still being filled in at this point: take the configs / params through `memo`, which hands out the
        # very objects the copied module tree gets)
        new = Engine.__new__(Engine)
        new._setup(weakref.ref(owner), copy.deepcopy(self.enc_cfg, memo), copy.deepcopy(self.dec_cfg, memo),
                   copy.deepcopy(orig.params, memo))
        return new

    def close(self):
        """Drop captured decode sessions (hipGraphs + their private pool) now."""
        self._decode_sessions.clear()

    # ------------------------------------------------------------------------------------------ setup
    def prepare(self, device):
        """(Re)build the flat storage if the module tree / device changed (e.g. after `.to(device)` or the
        embedding aliasing of train_gen.py:293) and make sure the bf16 shadow weights are current."""
        if device.type != "cuda":
            raise GstvdError("gst_visdial_amd runs on MI355X only; tensors are on %s (no CPU path)" % device)
        ops.set_device(device)
        gen = self.model.decoder.decoder
        topo = (id(gen.bert.embeddings), id(self.model.encoder.bert_pretrained.bert.embeddings), id(gen.lm_head.decoder.weight))
        if self.flat is None or self.flat.topo != topo:
            self.flat = FlatParams(self.model, self.precision)
            self.flat.topo = topo
            self.flat.stale_guard = lambda: self.pipe is not None and bool(getattr(self.pipe, "master_stale", False))
        if self.flat.is_materialized() and self.flat.device != device:
            # the parameters live (as views of the flat buffer) on one device and the inputs arrive on another: a replica of
            # nn.DataParallel(model, [0, 1, ...]) (train_gen.py:295, README.md:89) -- replicas would share this one engine
            raise GstvdError(
                "inputs on %s but the model's parameters are on %s: gst_visdial_amd runs one process per GPU.  "
                "nn.DataParallel(model, [0]) is fine; for N GPUs launch N ranks (torchrun --nproc-per-node N) and attach "
                "pipeline.BackwardPipeline for the gradient all-reduce (INTEGRATION.md, 'Multi-GPU')" % (device, self.flat.device))
        if not self.flat.is_materialized():
            self.flat.materialize(device)
            self._bind_views()
            self._decode_sessions.clear()         # captured graphs address the old buffers
        if self.arena is None or self.arena.device != device:
            self._decode_sessions.clear()
            self.arena = Arena(device)
            seed = int(self.model.params.get("amd_seed", 0))
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and self.model.params.get("amd_seed_per_rank", True):
                # one mask stream per data-parallel rank, like the per-device generators of the reference's DataParallel
                # replicas (train_gen.py:295); params['amd_seed_per_rank'] = False restores one shared stream
                seed = ops.rank_seed(seed, dist.get_rank())
            self.rng = ops.Rng(device, seed=seed)
            self.anchor = torch.zeros(1, device=device, requires_grad=True)
            self.side, self.aux = device_streams(device)
            self.aux_busy = False
            self.colsums = ops.ColsumBatch(device)
            self.wgrads = ops.GemmGroup(device, a_km=True, b_km=True)
        self.flat.refresh_shadow()

    def _bind_views(self):
        f = self.flat
        wbuf = f.S if self.precision == "bf16" else f.P
        self.W, self.Pv, self.Gv = {}, {}, {}
        for name in f.slots:
            self.Pv[name] = f.view(f.P, name)
            self.Gv[name] = f.view(f.G, name)
            self.W[name] = f.view(wbuf, name)

    # ------------------------------------------------------------------------------------------ helpers
    def buf(self, M, N, dtype=None):
        return self.arena.alloc(M * N, dtype or self.adt, (M, N))

    def vec(self, n, dtype=torch.float32):
        return self.arena.alloc(n, dtype)

    def act(self, M, N):
        return Act(self.buf(M, N), M, N)

    def site(self, label=None, p=0.0, kind=None, shape=None, **extra):
        """Next dropout site of the step.  Sites are numbered in issue order; `site_log[label]` keeps what a checker needs to
        regenerate the site's keep mask with gstvd_dropout_mask (tests: the oracle is run with exactly these masks)."""
        self._site += 1
        if label is not None and self.train:
            if label in self.site_log:
                raise GstvdError("internal: dropout site label used twice in one step: " + label)
            self.site_log[label] = dict(site=self._site, p=float(p), kind=kind, shape=shape, **extra)
        return self._site

    def grad_slot(self, name):
        g = self.Gv[name]
        off = self.flat.slots[name][0]
        acc = self.accumulate or (off in self.written)
        self.written.add(off)
        return g, acc

    def push(self, fn):
        if self.rec:
            sc = ops.Profiler.scope
            if sc is not None and ops.Profiler.active is not None:      # instrumented step: backward inherits the block label
                def scoped(fn=fn, sc=sc):
                    prev, ops.Profiler.scope = ops.Profiler.scope, sc
                    try:
                        fn()
                    finally:
                        ops.Profiler.scope = prev
                self.tape.append((self.tag, scoped))
            else:
                self.tape.append((self.tag, fn))

    def mark(self, key):
        if self.rec and self.pipe is not None:
            off = self.flat.marks[key]
            self.tape.append(("t", lambda: self._hook(off)))

    def _hook(self, off):
        """Backward reached flat offset `off`: every gradient at offset >= off has been produced or queued."""
        if self.pipe is not None and self.pipe.ready(off):
            self._emit(off)

    def _early_final_wgrads(self):
        """Single-slice pipeline with the fused update (the N = 1 default): every weight-gradient GEMM of the step is queued once
        backward reaches the encoder's text embedding -- what is left is that embedding's backward (scatter-adds, 33 us), and the
        column reductions of the LayerNorm / bias gradients (47 us): inputs of the REMAINDER AdamW pass, not of the grouped
        launch.  So the grouped launch starts here on the auxiliary stream and those two run beside it instead of in front of it
        (engine.EARLY_WGRAD = 0: the launch waits for them, as in rounds 4-5a; profiles/r05_early_wgrad_ab.txt)."""
        p = self.pipe
        if not self.use_streams or p is None or p.hi is None or p.slices or p.hi != self.flat.n_live or p.fuse_handle() is None:
            return
        if any(self.wgrads.pending_into(self.Gv[n]) for n in ("emb.word", "emb.pos", "emb.tt", "emb.tte") if n in self.Gv):
            return      # a queued GEMM writes a table the embedding's backward is about to add into (tied LM head): keep the order
        for src in (self.main, self.side):
            ev = torch.cuda.Event()
            ev.record(src)
            self.aux.wait_event(ev)
        with torch.cuda.stream(self.aux):
            self._early_fused = self.wgrads.flush(fuse=p.fuse_handle())
        self._early = True
        self.aux_busy = True

    def _flush_wgrads(self):
        """The slice's queued weight gradients: -> (fused, direct).  N = 1: one launch that also applies AdamW (`fused` offsets).
        N > 1 with a bf16 payload: the launch writes the payload itself (`direct` ranges, pipeline.bf16_target)."""
        tgt = self.pipe.bf16_target()
        if tgt is not None:
            return (), self.wgrads.flush_direct_bf16(*tgt)
        return self.wgrads.flush(fuse=self.pipe.fuse_handle()), ()

    def _emit(self, off):
        """Hand the finished slice [off, pipe.hi) to the backward pipeline on the auxiliary stream."""
        if getattr(self.pipe, "segmenter", None) is not None:
            # Segmented capture (graph.SegmentedStep: the fall-back when a whole-step capture WITH its collectives is refused): the
            # slice's collective is issued eagerly BETWEEN two captured graphs, so the capture is cut here.  A capture can only end
            # with every forked stream joined into its origin: the vision stream joins, the slice's weight gradients and column
            # reductions run on the main stream (beside the backward chain they are zero-sum anyway), run_slice cuts, and the vision
            # stream is forked into the next capture at once -- its next tape entry may come before the next recorded dependency,
            # and a launch on a stream outside the capture would run at capture time instead of being captured.
            if self.use_streams:
                self._wait("t", "v")
                if self.aux_busy:
                    ev = torch.cuda.Event()
                    ev.record(self.aux)
                    self.main.wait_event(ev)
                    self.aux_busy = False
            fused, direct = self._flush_wgrads()
            self.colsums.flush()
            self.pipe.run_slice(off, self.pipe.hi, fused=fused, direct=direct)
            if self.use_streams:
                self._wait("v", "t")
            return
        for src in ([self.main, self.side] if self.use_streams else [self.main]):
            ev = torch.cuda.Event()
            ev.record(src)
            self.aux.wait_event(ev)
        if off == 0 and self.use_streams and getattr(self, "_early", False):
            # the grouped launch is already running (see _early_final_wgrads); a GEMM queued after it (none in this model) goes plain
            self._early = False
            fused = self._early_fused
            with torch.cuda.stream(self.aux):
                self.wgrads.flush()
            self.colsums.flush()
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.aux.wait_event(ev)
            with torch.cuda.stream(self.aux):
                self.pipe.run_slice(off, self.pipe.hi, fused=fused)
            self.aux_busy = True
            return
        if off == 0 and self.use_streams:
            # the last slice, after backward's last kernel: the main stream has nothing left to do, so the slice's column
            # reductions run there, beside its grouped weight-gradient launch on the auxiliary stream (which then waits for them).
            # (Round 3 measured cutting this slice into 2-6 parts so that AdamW of part j runs beside the weight gradients of part
            # j+1, also with the weight-gradient launch confined to 160-224 CUs: 13.73-14.01 ms against 13.72 ms, resp. +0.3 ms --
            # the two full-chip kernels do not share the chip profitably; profiles/r03_chunk_sweep.txt.  Not kept.)
            with torch.cuda.stream(self.aux):
                fused, direct = self._flush_wgrads()
            # (issuing these reductions from the vision stream instead does not let the grouped launch start earlier:
            # 12.30 / 12.34 vs 12.31 / 12.35 ms, tools/r05_s14.sh)
            self.colsums.flush()
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.aux.wait_event(ev)
            with torch.cuda.stream(self.aux):
                self.pipe.run_slice(off, self.pipe.hi, fused=fused, direct=direct)
        else:
            with torch.cuda.stream(self.aux):
                fused, direct = self._flush_wgrads()
                self.colsums.flush()
                self.pipe.run_slice(off, self.pipe.hi, fused=fused, direct=direct)
        self.aux_busy = True

    # -- two HIP streams: the vision stream's skinny (M = B*37) kernels run beside the text stream's --------------
    class _On(object):
        def __init__(self, eng, tag):
            self.eng, self.tag, self.ctx = eng, tag, None

        def __enter__(self):
            self.prev = self.eng.tag
            self.eng.tag = self.tag
            if self.eng.use_streams and self.tag == "v":
                self.ctx = torch.cuda.stream(self.eng.side)
                self.ctx.__enter__()

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)
            self.eng.tag = self.prev

    def on(self, tag):
        return Engine._On(self, tag)

    def _flush_aux(self):
        """Run the queued (decoder + LM head) weight-gradient group and column reductions on a third stream so they
        overlap the encoder's backward chain; joined back before the final flush."""
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.aux.wait_event(ev)
        with torch.cuda.stream(self.aux):
            self.wgrads.flush()
            self.colsums.flush()
        self.aux_busy = True

    def _stream_of(self, tag):
        return self.side if tag == "v" else self.main

    def _wait(self, waiter, waitee):
        """Stream `waiter` waits for everything queued so far on stream `waitee`."""
        ev = torch.cuda.Event()
        ev.record(self._stream_of(waitee))
        self._stream_of(waiter).wait_event(ev)

    def sync(self, waiter, waitee):
        """Forward dependency (ops on `waiter` after this point read results of `waitee`); backward mirrors it."""
        if not self.use_streams:
            return
        self._wait(waiter, waitee)
        if self.rec:
            self.tape.append(("sync", (waitee, waiter)))

    # ------------------------------------------------------------------------------------------ ops
    def lin(self, x, w, b, N, K, gelu=False, need_dx=True):
        y = self.act(x.M, N)
        if gelu:
            u = self.buf(x.M, N)
            ops.gemm(x.t, self.W[w], y.t, x.M, N, K, bias=self.Pv[b], aux=u, epi=EPI_GELU)
            y.gelu_aux = u
        else:
            ops.gemm(x.t, self.W[w], y.t, x.M, N, K, bias=self.Pv[b])
        y.prod = (x, w, K, need_dx)
        self.push(lambda: self._lin_bwd(x, y, w, b, N, K, need_dx))
        return y

    def ln_lin(self, x, res, g, b, H, p_pre, bias_name, eps, w, wb, N, gelu=False, mark=None):
        """y = LN(drop(x) + res); out = Linear_w(y) (+ GELU): ONE launch when the rows are few enough that every launch is
        latency bound (gstvd_gemm_ln_fwd; the decoder's ln1 -> cross-attention query, ln2 -> FFN up, ln3 -> next layer's
        QKV), else the two kernels.  `mark`: a pipeline watermark that belongs between the two (the LayerNorm's parameters
        sit in the previous layer's slice of the flat buffer, the Linear's in the next one's).  Returns (y, out)."""
        M = x.M
        if not (self.fuse_ln and res is not None and ops.gemm_ln_ok(M, N, H, self.adt)):
            y = self.ln(x, res, g, b, H, p_pre, bias_name, eps)
            if mark is not None:
                self.mark(mark)
            return y, self.lin(y, w, wb, N, H, gelu=gelu)
        y, out = self.act(M, H), self.act(M, N)
        kw = dict(mode=LN_RESID, dtype=ops.dt(x.t), M=M, H=H, gamma=self.Pv[g], beta=self.Pv[b],
                  mean=self.vec(M), rstd=self.vec(M), eps=eps, x=x.t, res=res.t, y=y.t,
                  p_pre=p_pre if self.train else 0.0, site_pre=self.site(g[:-2], p_pre, "rows", (M, H)), rng=self.rng)
        aux = self.buf(M, N) if gelu else None
        ops.gemm_ln_fwd(kw, self.W[w], out.t, N, bias=self.Pv[wb], aux=aux, epi=EPI_GELU if gelu else 0)
        out.gelu_aux = aux
        out.prod = (y, w, H, True)
        self.push(lambda: self._ln_bwd(kw, x, res, y, g, b, H, bias_name))
        if mark is not None:
            self.mark(mark)
        self.push(lambda: self._lin_bwd(y, out, w, wb, N, H, True))
        return y, out

    def _lin_bwd(self, x, y, w, b, N, K, need_dx):
        dy, M = y.g, x.M      # for a GELU output y.g already holds d(pre-activation): its producer applied gelu'
        gw, acc = self.grad_slot(w)
        if not y.bias_done and self.wgrads.colsum_capable(dy):
            # the bias gradient (column sums of dY) comes out of the weight-gradient launch itself: its producer waves sum the
            # dY tiles they stage anyway
            gb, accb = self.grad_slot(b)
            self.wgrads.add(dy, x.t, gw, N, K, M, acc, colsum_out=gb, colsum_acc=accb)
        else:
            self.wgrads.add(dy, x.t, gw, N, K, M, acc)   # deferred: all weight-gradient GEMMs run as one grouped launch
            if not y.bias_done:
                gb, accb = self.grad_slot(b)
                scratch = self.vec(((M + 63) // 64) * N)
                self.colsums.add_slabs(dy, M, N, scratch, gb, accb)
        if need_dx and not y.dgrad_done:          # (dgrad_done: the LayerNorm behind y ran it with its own backward, _ln_bwd)
            add = x.g
            if x.g is None:
                x.g = self.buf(M, K)
            ops.gemm(dy, self.W[w], x.g, M, K, N, b_km=True, addend=add, aux=x.gelu_aux,
                     epi=EPI_DGELU if x.gelu_aux is not None else 0)

    def ln(self, x, res, g, b, H, p_pre, bias_name, eps=1e-12):
        M = x.M
        y = self.act(M, H)
        kw = dict(mode=LN_RESID, dtype=ops.dt(x.t), M=M, H=H, gamma=self.Pv[g], beta=self.Pv[b],
                  mean=self.vec(M), rstd=self.vec(M), eps=eps, x=x.t, res=res.t if res is not None else None, y=y.t,
                  p_pre=p_pre if self.train else 0.0, site_pre=self.site(g[:-2], p_pre, "rows", (M, H)), rng=self.rng)
        ops.ln_fwd(**kw)
        self.push(lambda: self._ln_bwd(kw, x, res, y, g, b, H, bias_name))
        return y

    def _colsums(self, partial, nblk, H, names):
        """Queue the reduction of LN-backward partials into up to three gradient slots (None = skip)."""
        outs, accs = [], []
        for n in names:
            if n is None:
                outs.append(None); accs.append(False)
            else:
                gv, a = self.grad_slot(n)
                outs.append(gv); accs.append(a)
        self.colsums.add(partial, outs, nblk, 3 * H, H, 3, accs)

    def _ln_bwd(self, kw, x, res, y, g, b, H, bias_name):
        M = x.M
        prod = x.prod
        # (prod[0] must be a third tensor: for LN(Linear(r) + r) the input gradient of the Linear and the residual gradient are the
        # SAME tensor's gradient -- the fused launch would hand one uninitialised buffer to both its `dres` column tile and its
        # dgrad addend/C, an intra-kernel race; such a site takes the two-launch path, which accumulates correctly)
        if (self.fuse_ln and prod is not None and prod[3] and res is not None and prod[0].N == prod[2]
                and prod[0] is not res and prod[0] is not x and ops.gemm_ln_ok(M, prod[2], H, self.adt)):
            # the LayerNorm's backward and the input gradient of the Linear that produced its input, one launch
            # (gstvd_gemm_ln_bwd): dx never travels through HBM between the two, one dependent launch less per sub-layer
            xin, w, Kin, _ = prod
            R = ops.gemm_ln_rows()
            nblk = (M + R - 1) // R
            partial = self.arena.alloc(nblk * 3 * H, torch.float32)
            if res.g is not None:
                raise GstvdError("internal: residual gradient written twice")
            res.g = self.buf(M, H)
            x.g = self.buf(M, H)
            add = xin.g
            if xin.g is None:
                xin.g = self.buf(M, Kin)
            ops.gemm_ln_bwd(kw, y.g, partial, nblk, self.W[w], xin.g, Kin, dres=res.g, dx=x.g, addend=add, aux=xin.gelu_aux,
                            epi=EPI_DGELU if xin.gelu_aux is not None else 0)
            x.dgrad_done = True
            self._colsums(partial, nblk, H, [g, b, bias_name])
            x.bias_done = bias_name is not None
            return
        nblk = ops.ln_bwd_blocks(M, H, LN_RESID)
        partial = self.arena.alloc(nblk * 3 * H, torch.float32)
        if res is not None:
            if res.g is not None:
                raise GstvdError("internal: residual gradient written twice")
            res.g = self.buf(M, H)
        x.g = self.buf(M, H)
        ops.ln_bwd(kw, y.g, partial, dres=res.g if res is not None else None, dx=x.g, nblk=nblk)
        self._colsums(partial, nblk, H, [g, b, bias_name])
        x.bias_done = bias_name is not None

    def embed(self, prefix, ids, segs, Bn, T, cfg, pos_offset=0, label=None, p_drop=None):
        M, H = Bn * T, cfg.hidden_size
        if p_drop is None:
            p_drop = cfg.hidden_dropout_prob
        y = self.act(M, H)
        kw = dict(mode=LN_EMBED, dtype=ops.dt(y.t), M=M, H=H, gamma=self.Pv[prefix + ".ln.w"], beta=self.Pv[prefix + ".ln.b"],
                  mean=self.vec(M), rstd=self.vec(M), eps=1e-12, y=y.t, ids=ids, segs=segs, T=T,
                  type_vocab=cfg.type_vocab_size, word=self.Pv[prefix + ".word"], pos=self.Pv[prefix + ".pos"],
                  tt=self.Pv[prefix + ".tt"], tt_ext=self.Pv[prefix + ".tte"],
                  p_post=p_drop if self.train else 0.0, site_post=self.site(label, p_drop, "rows", (M, H)), rng=self.rng,
                  pos_offset=pos_offset)
        ops.ln_fwd(**kw)
        self.push(lambda: self._embed_bwd(kw, prefix, y, M, H))
        return y

    def _embed_bwd(self, kw, prefix, y, M, H):
        tabs, fresh = [], []
        for n in (".word", ".pos", ".tt", ".tte"):
            gv, acc = self.grad_slot(prefix + n)
            if not acc:
                fresh.append(n)
            elif self.wgrads.pending_into(gv):
                # the LM head tied to this table (FlatParams aliases lm.w onto it when train_gen.py:293 was not applied) has its
                # weight-gradient GEMM queued: run the queue now, in order, WITHOUT the fused update -- the scatter-add below is a
                # second contribution to the same slot, and a GEMM launched later would overwrite it
                self.wgrads.flush()
            tabs.append(gv)
        if len(fresh) == 4:
            # the four tables are laid out back to back (FlatParams.emb; alignment gaps and the word table's row padding belong to
            # no parameter): one fill instead of four dependent ones on the backward chain
            lo = self.flat.slots[prefix + ".word"][0]
            hi = self.flat.slots[prefix + ".tte"][0] + tabs[3].numel()
            ok = self._emb_span_ok.get((prefix, lo, hi))
            if ok is None:
                # no OTHER parameter's slot may start inside the span (its finished gradient would be wiped): checked against the
                # layout itself, once per layout, instead of against a size bound with a whole padded table of slack
                mine = set(prefix + n for n in (".word", ".pos", ".tt", ".tte"))
                own = set(self.flat.slots[n][0] for n in mine)
                ok = hi >= lo and all(not (lo <= off < hi) or off in own for name, (off, _) in self.flat.slots.items() if name not in mine)
                self._emb_span_ok[(prefix, lo, hi)] = ok
            if not ok:
                raise GstvdError("internal: embedding tables are not contiguous in the flat buffer")
            self.flat.G[lo:hi].zero_()
        else:
            for n, gv in zip((".word", ".pos", ".tt", ".tte"), tabs):
                if n in fresh:
                    gv.zero_()
        nblk = ops.ln_bwd_blocks(M)
        partial = self.arena.alloc(nblk * 4 * H, torch.float32)
        ops.ln_bwd(kw, y.g, partial, dword=tabs[0], dpos=tabs[1], dtt=tabs[2], dtt_ext=tabs[3])
        gw, aw = self.grad_slot(prefix + ".ln.w")
        gb, ab = self.grad_slot(prefix + ".ln.b")
        tt = tabs[2]                                   # zeroed above unless accumulating: rows 0/1 always add
        self.colsums.add(partial, (gw, gb, tt[0]), nblk, 4 * H, H, 3, (aw, ab, True))
        if tt.shape[0] > 1:
            self.colsums.add(partial[3 * H:], (tt[1], None, None), nblk, 4 * H, H, 1, (True, False, False))

    def img_embed(self, x, loc, cfg):
        M, H = x.M, cfg.v_hidden_size
        y = self.act(M, H)
        kw = dict(mode=LN_IMAGE, dtype=ops.dt(x.t), M=M, H=H, gamma=self.Pv["vemb.ln.w"], beta=self.Pv["vemb.ln.b"],
                  mean=self.vec(M), rstd=self.vec(M), eps=1e-12, x=x.t, y=y.t, loc=loc, w_loc=self.Pv["vemb.loc.w"],
                  b_loc=self.Pv["vemb.loc.b"], p_post=cfg.hidden_dropout_prob if self.train else 0.0,
                  site_post=self.site("vemb", cfg.hidden_dropout_prob, "rows", (M, H)), rng=self.rng)
        ops.ln_fwd(**kw)
        self.push(lambda: self._img_embed_bwd(kw, x, y, loc, M, H))
        return y

    def _img_embed_bwd(self, kw, x, y, loc, M, H):
        nblk = ops.ln_bwd_blocks(M)
        partial = self.arena.alloc(nblk * 3 * H, torch.float32)
        x.g = self.buf(M, H)
        ops.ln_bwd(kw, y.g, partial, dres=x.g)
        self._colsums(partial, nblk, H, ["vemb.ln.w", "vemb.ln.b", "vemb.loc.b"])
        self._colsums(partial, nblk, H, [None, None, "vemb.img.b"])
        x.bias_done = True
        gw, acc = self.grad_slot("vemb.loc.w")
        ops.locgrad(x.g, loc, M, H, gw, acc)

    def attn(self, q, k, v, Bn, nh, Lq, Lk, d, key_mask, causal, neg, p, kv_group=1, kv_bstride=0, label=None):
        (qa, qc), (ka, kc), (va, vc) = q, k, v
        Hh = nh * d
        o = self.act(Bn * Lq, Hh)
        lse = self.vec(Bn * nh * Lq)
        # text self-attention in training: forward leaves the keep bits of its dropout draws (8 KB per head) for the one-pass
        # backward, which then reads a bit instead of hashing every draw a second time
        nbits = ops.attn_keep_bits_shape(Bn, nh, Lq, Lk, d, self.adt, causal, p if (self.train and self.rec) else 0.0) if kv_group == 1 else 0
        bits = self.arena.alloc(nbits, torch.int64) if nbits else None
        a = ops.attn_desc(qa.t[:, qc:qc + Hh], ka.t[:, kc:kc + Hh], va.t[:, vc:vc + Hh], o.t, lse, key_mask, Bn, nh, Lq, Lk, d,
                          causal=causal, mask_neg=neg, drop_p=p if self.train else 0.0,
                          site=self.site(label, p, "attn", (Bn, nh, Lq, (Lk + 3) // 4 * 4), Lk=Lk), rng=self.rng,
                          kv_group=kv_group, kv_bstride=kv_bstride, drop_bits=bits)
        ops.attn_fwd(a)
        self.push(lambda: self._attn_bwd(a, q, k, v, o, Bn, nh, Lq, Hh))
        return o

    def _attn_bwd(self, a, q, k, v, o, Bn, nh, Lq, Hh):
        gs = []
        for (act, c) in (q, k, v):
            if act.g is None:
                act.g = self.buf(act.M, act.N)
            gs.append(act.g[:, c:c + Hh])
        delta = self.vec(Bn * nh * Lq)
        ops.attn_bwd(a, o.g, gs[0], gs[1], gs[2], delta)

    # ------------------------------------------------------------------------------------------ blocks
    def self_block(self, p, x, Bn, L, H, nh, key_mask, pa, ph, causal=False, inter=0):
        """QKV -> attention -> output dense -> dropout -> LN(+x)   (vilbert_dialog.py:380-431).  With `inter` the FFN's first
        Linear (+ GELU) is issued with the LayerNorm (ln_lin: one launch at the vision stream's row counts, the two kernels at
        the text stream's) and (x1, a) is returned for ffn_block(..., a=a)."""
        qkv = self.lin(x, p + ".qkv.w", p + ".qkv.b", 3 * H, H)
        ctx = self.attn((qkv, 0), (qkv, H), (qkv, 2 * H), Bn, nh, L, L, H // nh, key_mask, causal, -10000.0, pa, label=p + ".attn")
        ao = self.lin(ctx, p + ".ao.w", p + ".ao.b", H, H)
        if inter:
            return self.ln_lin(ao, x, p + ".ln1.w", p + ".ln1.b", H, ph, p + ".ao.b", 1e-12, p + ".fi.w", p + ".fi.b", inter, gelu=True)
        return self.ln(ao, x, p + ".ln1.w", p + ".ln1.b", H, ph, p + ".ao.b")

    def ffn_block(self, p, x, H, inter, ph, ti=".fi", to=".fo", tl=".ln2", a=None):
        """dense+GELU -> dense -> dropout -> LN(+x)   (vilbert_dialog.py:445-462); `a`: the first Linear's output when the caller
        issued it with the LayerNorm in front (ln_lin)."""
        if a is None:
            a = self.lin(x, p + ti + ".w", p + ti + ".b", inter, H, gelu=True)
        fo = self.lin(a, p + to + ".w", p + to + ".b", H, inter)
        return self.ln(fo, x, p + tl + ".w", p + tl + ".b", H, ph, p + to + ".b")

    def conn_layer(self, p, xv, xt, Bn, R, T, I):
        """BertConnectionLayer (vilbert_dialog.py:646-773): stream 1 = vision, 2 = text; ctx1 (text queries over
        vision keys) feeds the text branch, ctx2 the vision branch."""
        c = self.enc_cfg
        H, Hv, Hb, nh = c.hidden_size, c.v_hidden_size, c.bi_hidden_size, c.bi_num_attention_heads
        d = Hb // nh
        prev_scope, ops.Profiler.scope = ops.Profiler.scope, "coattn"
        try:
            return self._conn_layer(p, xv, xt, Bn, R, T, I, c, H, Hv, Hb, nh, d)
        finally:
            ops.Profiler.scope = prev_scope

    def _conn_layer(self, p, xv, xt, Bn, R, T, I, c, H, Hv, Hb, nh, d):
        with self.on("v"):
            qkv1 = self.lin(xv, p + ".qkv1.w", p + ".qkv1.b", 3 * Hb, Hv)
        qkv2 = self.lin(xt, p + ".qkv2.w", p + ".qkv2.b", 3 * Hb, H)
        self.sync("t", "v")
        self.sync("v", "t")
        ctx1 = self.attn((qkv2, 0), (qkv1, Hb), (qkv1, 2 * Hb), Bn, nh, T, R, d, I["vmask"], False, -10000.0,
                         c.v_attention_probs_dropout_prob, label=p + ".attn1")
        with self.on("v"):
            ctx2 = self.attn((qkv1, 0), (qkv2, Hb), (qkv2, 2 * Hb), Bn, nh, R, T, d, I["tmask"], False, -10000.0,
                             c.attention_probs_dropout_prob, label=p + ".attn2")
            hv = self.lin(ctx2, p + ".d1.w", p + ".d1.b", Hv, Hb)
            av, a = self.ln_lin(hv, xv, p + ".ln1.w", p + ".ln1.b", Hv, c.v_hidden_dropout_prob, p + ".d1.b", 1e-12,
                                p + ".vfi.w", p + ".vfi.b", c.v_intermediate_size, gelu=True)
            ov = self.ffn_block(p, av, Hv, c.v_intermediate_size, c.v_hidden_dropout_prob, ".vfi", ".vfo", ".vln", a=a)
        ht = self.lin(ctx1, p + ".d2.w", p + ".d2.b", H, Hb)
        at, a = self.ln_lin(ht, xt, p + ".ln2.w", p + ".ln2.b", H, c.hidden_dropout_prob, p + ".d2.b", 1e-12,
                            p + ".tfi.w", p + ".tfi.b", c.intermediate_size, gelu=True)
        ot = self.ffn_block(p, at, H, c.intermediate_size, c.hidden_dropout_prob, ".tfi", ".tfo", ".tln", a=a)
        return ov, ot

    def encoder(self, I):
        """BertModel.forward for enc_dec (vilbert_dialog.py:1325-1407); poolers / cls heads are dead and skipped."""
        c = self.enc_cfg
        Bn, T, R = I["B"], I["T"], I["R"]
        xt = self.embed("emb", I["ids"], I["segs"], Bn, T, c, label="emb.enc")
        if self.rec and self.pipe is not None and EARLY_WGRAD:
            self.tape.append(("t", self._early_final_wgrads))       # backward: runs just before this embedding's backward
        f = Act(I["feats"], Bn * R, c.v_feature_size)
        self.sync("v", "t")                   # fork: the vision stream starts once the inputs are staged
        with self.on("v"):
            g0 = self.lin(f, "vemb.img.w", "vemb.img.b", c.v_hidden_size, c.v_feature_size, need_dx=I["feats_grad"])
            xv = self.img_embed(g0, I["loc"], c)
        I["feats_act"] = f
        for kind, i in encoder_schedule(c):
            self.mark((kind, i))
            if kind == "t":
                p = "t%d" % i
                x1, a = self.self_block(p, xt, Bn, T, c.hidden_size, c.num_attention_heads, I["tmask"],
                                        c.attention_probs_dropout_prob, c.hidden_dropout_prob, inter=c.intermediate_size)
                xt = self.ffn_block(p, x1, c.hidden_size, c.intermediate_size, c.hidden_dropout_prob, a=a)
            elif kind == "v":
                p = "v%d" % i
                with self.on("v"):
                    x1, a = self.self_block(p, xv, Bn, R, c.v_hidden_size, c.v_num_attention_heads, I["vmask"],
                                            c.v_attention_probs_dropout_prob, c.v_hidden_dropout_prob, inter=c.v_intermediate_size)
                    xv = self.ffn_block(p, x1, c.v_hidden_size, c.v_intermediate_size, c.v_hidden_dropout_prob, a=a)
            else:
                xv, xt = self.conn_layer("c%d" % i, xv, xt, Bn, R, T, I)
        self.sync("t", "v")                   # join before VLFusion
        return xt, xv

    def fusion(self, xt, xv, I):
        """VLFusion (visual_dialog_model.py:131-135): cat(fc_v(h_v), fc_l(h_t)) along the sequence, dropout 0.1."""
        c = self.enc_cfg
        Bn, T, R = I["B"], I["T"], I["R"]
        S, H, Hv = R + T, c.hidden_size, c.v_hidden_size
        self.mark("vlf")
        enc = self.act(Bn * S, H)
        p = 0.1 if self.train else 0.0
        sv, st = self.site("vlf.v", 0.1, "rows", (Bn * R, H)), self.site("vlf.l", 0.1, "rows", (Bn * T, H))
        e3 = enc.t.view(Bn, S, H)
        ops.gemm(xv.t, self.W["vlf.v.w"], e3[:, :R], R, H, Hv, bias=self.Pv["vlf.v.b"], batch=Bn, sA=R * Hv, sC=S * H,
                 lda=Hv, ldc=H, drop_p=p, site=sv, rng=self.rng)
        ops.gemm(xt.t, self.W["vlf.l.w"], e3[:, R:], T, H, H, bias=self.Pv["vlf.l.b"], batch=Bn, sA=T * H, sC=S * H,
                 lda=H, ldc=H, drop_p=p, site=st, rng=self.rng)
        self.push(lambda: self._fusion_bwd(enc, xt, xv, Bn, R, T, H, Hv, p, sv, st))
        return enc

    def _fusion_bwd(self, enc, xt, xv, Bn, R, T, H, Hv, p, sv, st):
        yv, yt = self.act(Bn * R, H), self.act(Bn * T, H)
        yv.g, yt.g = yv.t, yt.t
        ops.vl_split(enc.g, Bn, R, T, H, yv.g, yt.g, p, sv, st, self.rng)
        self._lin_bwd(xv, yv, "vlf.v.w", "vlf.v.b", H, Hv, True)
        self._lin_bwd(xt, yt, "vlf.l.w", "vlf.l.b", H, H, True)

    def decoder(self, enc, I, kv=None, kv_group=1):
        """BertGenerationEncoder + HF BertEncoder (self-attn -> cross-attn -> FFN, post-LN) + LM head.
        kv_group > 1 (inference): `kv_group` consecutive decoder rows attend to the same encoder row."""
        c = self.dec_cfg
        Bn, U, S = I["B"] * kv_group, I["U"], I["R"] + I["T"]
        H, nh, L = c.hidden_size, c.num_attention_heads, c.num_hidden_layers
        d = H // nh
        eps = c.layer_norm_eps
        self.mark("dec")
        if self.rec and self.use_streams and self.pipe is None:
            self.tape.append(("t", self._flush_aux))      # backward: the decoder's gradients are complete here
        kv_on_side = False
        if kv is None:
            if self.use_streams:
                # the cross-attention K/V of all layers (one 4688 x 18432 x 768 GEMM, 0.19 ms; its input gradient as much) goes
                # to the side stream, idle since the encoder joined: it runs beside the decoder's embedding and the first layer's
                # self-attention sub-layer, and in backward its input gradient beside what is left of the decoder's backward
                self.sync("v", "t")
                with self.on("v"):
                    kv = self.lin(enc, "dec.ckv.w", "dec.ckv.b", 2 * L * H, H)
                kv_on_side = True
            else:
                kv = self.lin(enc, "dec.ckv.w", "dec.ckv.b", 2 * L * H, H)
        shared = self.flat.dec_emb is self.flat.enc_emb
        # the embedding MODULE's own dropout: with the shared module of train_gen.py:293 that is the ENCODER config's
        # hidden_dropout_prob (vilbert_dialog.py:321), whoever calls it (tests/golden/tiny_train_dropout.npz)
        y = self.embed("emb" if shared else "demb", I["dec_ids"], None, Bn, U, c, label="emb.dec",
                       p_drop=self.enc_cfg.hidden_dropout_prob if shared else None)
        # Each of the three LayerNorms of a layer is issued together with the Linear that reads it (ln_lin: one launch at the
        # decoder's row counts) -- ln3 with the NEXT layer's QKV projection, hence `pend`; its backward runs with the input
        # gradient of the Linear in front of it (_ln_bwd).  8 launches per layer and direction instead of 11.
        pend = None                                   # (fo, y2, prefix) of the previous layer: its ln3 is still to be issued
        for i in range(L):
            p = "d%d" % i
            if pend is None:
                self.mark(("d", i))
                qkv = self.lin(y, p + ".qkv.w", p + ".qkv.b", 3 * H, H)
            else:
                fo_, y2_, pp = pend
                y, qkv = self.ln_lin(fo_, y2_, pp + ".ln3.w", pp + ".ln3.b", H, c.hidden_dropout_prob, pp + ".fo.b", eps,
                                     p + ".qkv.w", p + ".qkv.b", 3 * H, mark=("d", i))
            ctx = self.attn((qkv, 0), (qkv, H), (qkv, 2 * H), Bn, nh, U, U, d, I["dmask"], True, -10000.0,
                            c.attention_probs_dropout_prob, label=p + ".attn")
            ao = self.lin(ctx, p + ".ao.w", p + ".ao.b", H, H)
            y1, q = self.ln_lin(ao, y, p + ".ln1.w", p + ".ln1.b", H, c.hidden_dropout_prob, p + ".ao.b", eps,
                                p + ".cq.w", p + ".cq.b", H)
            if kv_on_side and i == 0:
                self.sync("t", "v")               # the first cross-attention needs the K/V projection
            ctx = self.attn((q, 0), (kv, 2 * i * H), (kv, (2 * i + 1) * H), Bn, nh, U, S, d, I["emask"], False, -1e9,
                            c.attention_probs_dropout_prob, kv_group=kv_group, label=p + ".xattn")
            co = self.lin(ctx, p + ".co.w", p + ".co.b", H, H)
            y2, a = self.ln_lin(co, y1, p + ".ln2.w", p + ".ln2.b", H, c.hidden_dropout_prob, p + ".co.b", eps,
                                p + ".fi.w", p + ".fi.b", c.intermediate_size, gelu=True)
            fo = self.lin(a, p + ".fo.w", p + ".fo.b", H, c.intermediate_size)
            pend = (fo, y2, p)
        if pend is not None:
            fo_, y2_, pp = pend
            y = self.ln(fo_, y2_, pp + ".ln3.w", pp + ".ln3.b", H, c.hidden_dropout_prob, pp + ".fo.b", eps)
        self.mark("lm")
        logits = self.lin(y, "lm.w", "lm.b", self.flat.Vp, H)
        return y, logits

    # ------------------------------------------------------------------------------------------ step
    def _inputs(self, feats, loc, img_mask, ids, segs, att_mask, dec_ids, dec_mask):
        dev = ids.device
        Bn, T = ids.shape
        R = feats.shape[1]
        c = self.enc_cfg
        if T > c.max_position_embeddings:
            raise GstvdError("sequence length %d exceeds max_position_embeddings" % T)
        if segs is None:
            segs = torch.zeros_like(ids)
        if self._validate:   # the reference asserts on device every call (vilbert_dialog.py:339); here: first call / on request
            if int(segs.max()) >= c.type_vocab_size + 10 or int(segs.min()) < 0:
                raise GstvdError("segment id out of range")
            if int(ids.max()) >= c.vocab_size or int(ids.min()) < 0 or int(dec_ids.max()) >= c.vocab_size or int(dec_ids.min()) < 0:
                raise GstvdError("token id out of range")
            self._validate = bool(self.model.params.get("amd_validate_inputs", False))
        I = dict(B=Bn, T=T, R=R, U=dec_ids.shape[1])
        I["ids"], I["segs"] = ids.contiguous().view(-1), segs.contiguous().view(-1)
        tm = att_mask if att_mask is not None else torch.ones(Bn, T, device=dev)
        vm = img_mask if img_mask is not None else torch.ones(Bn, R, device=dev)
        I["tmask"], I["vmask"] = tm.float().contiguous(), vm.float().contiguous()
        em = self.arena.alloc(Bn * (R + T), torch.float32, (Bn, R + T))
        em[:, :R].copy_(I["vmask"])
        em[:, R:].copy_(I["tmask"])
        I["emask"] = em
        I["dmask"] = dec_mask.float().contiguous() if dec_mask is not None else None
        I["dec_ids"] = dec_ids.contiguous().view(-1)
        f2 = feats.reshape(Bn * R, feats.shape[-1])
        if self.adt == torch.float32:
            I["feats"] = f2.float().contiguous()
        else:
            fb = self.buf(Bn * R, f2.shape[1])
            ops.cast(f2.float().contiguous(), fb)
            I["feats"] = fb
        I["loc"] = loc.reshape(Bn * R, 5).float().contiguous()
        I["feats_grad"] = bool(feats.requires_grad and torch.is_grad_enabled())
        return I

    def _begin(self, device, record):
        self.prepare(device)
        self._last_decode = None          # the arena is rewound: a previous decode call's encoder states are about to be overwritten
        self.arena.reset()
        self.tape, self.rec = [], record
        self.tag = "t"
        self.main = torch.cuda.current_stream()
        self._site = 0
        self.site_log = {}
        self.train = bool(self.model.training)
        if self.train:
            self.rng.advance()

    def step(self, feats, loc, img_mask, ids, segs, att_mask, dec_ids, dec_mask, labels, loss_reduction=True):
        """EncoderDecoderModel.forward, train/eval branch -> (loss, logits)."""
        record = torch.is_grad_enabled()
        self._begin(ids.device, record)
        dc = self.dec_cfg
        if labels is None:    # visual_dialog_decoder.py:53-57: shift left, then mutate the caller's ids in place
            labels = dec_ids.new_zeros(dec_ids.shape)
            labels[:, :-1] = dec_ids[:, 1:].clone()
            dec_ids.masked_fill_(dec_ids == dc.eos_token_id, dc.pad_token_id)
        I = self._inputs(feats, loc, img_mask, ids, segs, att_mask, dec_ids, dec_mask)
        Bn, U, V = I["B"], I["U"], dc.vocab_size
        xt, xv = self.encoder(I)
        enc = self.fusion(xt, xv, I)
        y, logits = self.decoder(enc, I)
        Md = Bn * U
        lab = labels.contiguous().view(-1)
        row_loss, lse, stats = self.vec(Md), self.vec(Md), self.vec(4)
        ops.ce_fwd(logits.t, lab, Md, V, row_loss, lse, stats, ignore_index=dc.pad_token_id)
        st = dict(I=I, logits=logits, lab=lab, lse=lse, stats=stats, Md=Md, V=V, mean=bool(loss_reduction), tape=self.tape,
                  pad=dc.pad_token_id)
        self.last = dict(enc_t=xt, enc_v=xv, enc=enc, dec_hidden=y, logits=logits, lse=lse, row_loss=row_loss)
        lv = logits.t.view(Bn, U, self.flat.Vp)[:, :, :V]
        if record:
            loss_raw = stats[2] if loss_reduction else row_loss
            loss = _StepFn.apply(self.anchor, feats if I["feats_grad"] else None, self, st, loss_raw)
            return loss, lv
        loss = stats[2].clone() if loss_reduction else row_loss.clone()
        return loss, lv.to(torch.float32, copy=True)      # a copy: the arena is rewound by the next engine call

    def backward(self, st, gloss):
        """Replay the tape: fills the flat gradient buffer, assigns `.grad` views, returns d loss / d image features."""
        flat = self.flat
        if not st["mean"]:
            raise GstvdError("backward through loss_reduction=False is not supported (the reference never does it)")
        have = [p.grad is not None for p in flat.live]
        self.accumulate = any(have)
        if self.accumulate:
            for p, gv in zip(flat.live, flat.grad_views):
                if p.grad is None:
                    gv.zero_()
                elif p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
        self.written = set()
        self._early = False
        self.colsums.reset()
        self.wgrads.reset()
        logits = st["logits"]
        logits.g = self.buf(st["Md"], flat.Vp)
        gs = gloss.reshape(1).float().contiguous() if gloss is not None else None
        ops.ce_bwd(logits.t, st["lab"], st["lse"], st["stats"], gs, True, st["Md"], st["V"], logits.g, ignore_index=st["pad"])
        self.main, self.tag = torch.cuda.current_stream(), "t"
        if self.pipe is not None:
            self.pipe.begin()
        for tag, fn in reversed(st["tape"]):
            if tag == "sync":
                self._wait(*fn)
            elif tag == "v" and self.use_streams:
                with torch.cuda.stream(self.side):
                    fn()
            else:
                fn()
        if self.use_streams:
            self._wait("t", "v")
        if self.pipe is not None:
            if self.pipe.hi > 0:
                self._emit(0)
            tail = self.pipe.end()
            if tail is not None:                       # communication stream (all-reduce + AdamW of the slices): join it here
                self.main.wait_event(tail)
        if self.aux_busy:
            ev = torch.cuda.Event()
            ev.record(self.aux)
            self.main.wait_event(ev)
            self.aux_busy = False
        self.wgrads.flush()
        self.colsums.flush()
        stale = self.pipe.stale_grad_offsets() if self.pipe is not None else ()
        if stale:
            # the weight-gradient launch updated these weights in its epilogue and never stored dW: `.grad` stays None (a view
            # of the flat buffer would show an older step's bytes to gradient clipping / logging, and the next backward would
            # accumulate onto them); BackwardPipeline(keep_grads=True) materialises them
            for (p, off), gv in zip(flat.items, flat.grad_views):
                p.grad = None if off in stale else gv
        else:
            for p, gv in zip(flat.live, flat.grad_views):
                p.grad = gv
        fa = st["I"].get("feats_act")
        if st["I"]["feats_grad"] and fa is not None and fa.g is not None:
            return fa.g.float()
        return None

    # ------------------------------------------------------------------------------------------ candidate scoring
    @torch.no_grad()
    def score_candidates(self, feats, loc, img_mask, ids, segs, att_mask, dec_ids, dec_mask, group):
        """evaluate_gen.py:45-106 without the 100x redundancy: the `group` answer candidates of a dialog round share ONE
        encoder pass and ONE cross-attention K/V projection (the reference re-encodes the identical context per candidate).
        Encoder-side tensors have E rows (one per dialog round), decoder-side tensors E*group rows ordered
        [round, candidate].  Returns score[E*group] = sum_u [tgt != 0] log softmax(logits)[u, tgt], tgt = ids shifted left;
        the decoder input is the eos->pad masked copy, as visual_dialog_decoder.py:53-57 makes it."""
        dc = self.dec_cfg
        E, rows = ids.shape[0], dec_ids.shape[0]
        if rows != E * group:
            raise GstvdError("score_candidates: %d decoder rows for %d encoder rows x %d candidates" % (rows, E, group))
        self._begin(ids.device, False)
        self.train = False
        dec_in = dec_ids.masked_fill(dec_ids == dc.eos_token_id, dc.pad_token_id)
        I = self._inputs(feats, loc, img_mask, ids, segs, att_mask, dec_in, dec_mask)
        xt, xv = self.encoder(I)
        enc = self.fusion(xt, xv, I)
        L, H = dc.num_hidden_layers, dc.hidden_size
        kv = self.lin(enc, "dec.ckv.w", "dec.ckv.b", 2 * L * H, H)
        _, logits = self.decoder(enc, I, kv, kv_group=group)
        U, V = I["U"], dc.vocab_size
        Md = rows * U
        tgt = dec_ids.new_zeros(dec_ids.shape)
        tgt[:, :-1] = dec_ids[:, 1:]
        row_loss, lse, stats = self.vec(Md), self.vec(Md), self.vec(4)
        ops.ce_fwd(logits.t, tgt.contiguous().view(-1), Md, V, row_loss, lse, stats, ignore_index=dc.pad_token_id)
        scores = torch.empty(rows, dtype=torch.float32, device=ids.device)
        ops.answer_scores(logits.t, lse, dec_ids.contiguous(), rows, U, scores)
        return scores

    # ------------------------------------------------------------------------------------------ sampling decode
    def _decode_plan(self, ins, L0, max_seq_len):
        """Builds the two device programs of a decode call on the engine's arena: `encode()` (encoder, VLFusion, the
        cross-attention K/V of all decoder layers -- once per call) and `one_token(tok, t)` (ONE token per row through
        the decoder stack at position t, its self-attention K/V appended to the per-layer caches) -> fp32 logits [B, V].
        `ins` = (feats, loc, img_mask, ids, segs, att_mask, dec_ids) are the tensors the kernels read."""
        feats, loc, img_mask, ids, segs, att_mask, dec_ids = ins
        dc = self.dec_cfg
        st = {}

        def encode():
            self._begin(ids.device, False)
            self.train = False
            I = self._inputs(feats, loc, img_mask, ids, segs, att_mask, dec_ids, None)
            xt, xv = self.encoder(I)
            enc = self.fusion(xt, xv, I)
            Bn = I["B"]
            L, H = dc.num_hidden_layers, dc.hidden_size
            st["I"], st["Bn"], st["S"] = I, Bn, I["R"] + I["T"]
            st["kv"] = self.lin(enc, "dec.ckv.w", "dec.ckv.b", 2 * L * H, H)      # cross K/V of all layers, once
            Umax = L0 + max_seq_len
            st["Umax"] = Umax
            # per-layer cache of the fused Q|K|V rows, [B, Umax, 3H]: the QKV GEMM of position t writes its output rows straight
            # into cache[:, t] (row stride Umax*3H), attention reads K / V from the same rows -- no append copies
            st["QKVc"] = [Act(self.buf(Bn * Umax, 3 * H), Bn * Umax, 3 * H) for _ in range(L)]
            st["mark"] = self.arena.mark()

        def one_token(tok, t):
            I, Bn, S, kv, QKVc, Umax = st["I"], st["Bn"], st["S"], st["kv"], st["QKVc"], st["Umax"]
            V, Vp = dc.vocab_size, self.flat.Vp
            L, H, nh, eps = dc.num_hidden_layers, dc.hidden_size, dc.num_attention_heads, dc.layer_norm_eps
            d = H // nh
            prefix = "emb" if self.flat.dec_emb is self.flat.enc_emb else "demb"
            self.arena.rewind(st["mark"])
            y = self.embed(prefix, tok.contiguous(), None, Bn, 1, dc, pos_offset=t)
            # bf16: the three LayerNorms of a layer are folded into the Linears that read them (gstvd_gemv_ln) and the residual
            # adds into the epilogues of the Linears in front of them -- 8 launches per layer instead of 11.  `pre` = the
            # rows whose LayerNorm (parameters `lnp`) the next Linear still has to apply.
            fuse = (self.adt is torch.bfloat16 and Bn <= 16 and H <= 1024
                    and bool(self.model.params.get("amd_decode_fuse_ln", True)))       # (the switch exists for the parity test)
            I_ = dc.intermediate_size
            pre, lnp = None, None
            for i in range(L):
                p = "d%d" % i
                rows_t = QKVc[i].t.view(Bn, Umax, 3 * H)[:, t]                     # [Bn, 3H] view, row stride Umax * 3H
                if pre is None:
                    ops.gemm(y.t, self.W[p + ".qkv.w"], rows_t, Bn, 3 * H, H, bias=self.Pv[p + ".qkv.b"])
                else:
                    y = self.act(Bn, H)
                    ops.gemv_ln(pre.t, self.W[p + ".qkv.w"], rows_t, Bn, 3 * H, H, self.Pv[lnp + ".w"], self.Pv[lnp + ".b"], eps,
                                y_out=y.t, bias=self.Pv[p + ".qkv.b"])
                qkv = Act(rows_t, Bn, 3 * H)
                ctx = self.attn((qkv, 0), (QKVc[i], H), (QKVc[i], 2 * H), Bn, nh, 1, t + 1, d, None, False, -10000.0, 0.0,
                                kv_bstride=Umax)
                if fuse:
                    pre1, y1, q = self.act(Bn, H), self.act(Bn, H), self.act(Bn, H)
                    ops.gemm(ctx.t, self.W[p + ".ao.w"], pre1.t, Bn, H, H, bias=self.Pv[p + ".ao.b"], addend=y.t)
                    ops.gemv_ln(pre1.t, self.W[p + ".cq.w"], q.t, Bn, H, H, self.Pv[p + ".ln1.w"], self.Pv[p + ".ln1.b"], eps,
                                y_out=y1.t, bias=self.Pv[p + ".cq.b"])
                    ctx = self.attn((q, 0), (kv, 2 * i * H), (kv, (2 * i + 1) * H), Bn, nh, 1, S, d, I["emask"], False, -1e9, 0.0)
                    pre2, y2, a, aux = self.act(Bn, H), self.act(Bn, H), self.act(Bn, I_), self.buf(Bn, I_)
                    ops.gemm(ctx.t, self.W[p + ".co.w"], pre2.t, Bn, H, H, bias=self.Pv[p + ".co.b"], addend=y1.t)
                    ops.gemv_ln(pre2.t, self.W[p + ".fi.w"], a.t, Bn, I_, H, self.Pv[p + ".ln2.w"], self.Pv[p + ".ln2.b"], eps,
                                y_out=y2.t, bias=self.Pv[p + ".fi.b"], aux=aux, epi=EPI_GELU)
                    pre = self.act(Bn, H)
                    ops.gemm(a.t, self.W[p + ".fo.w"], pre.t, Bn, H, I_, bias=self.Pv[p + ".fo.b"], addend=y2.t)
                    lnp = p + ".ln3"
                    continue
                ao = self.lin(ctx, p + ".ao.w", p + ".ao.b", H, H)
                y1 = self.ln(ao, y, p + ".ln1.w", p + ".ln1.b", H, 0.0, None, eps)
                q = self.lin(y1, p + ".cq.w", p + ".cq.b", H, H)
                ctx = self.attn((q, 0), (kv, 2 * i * H), (kv, (2 * i + 1) * H), Bn, nh, 1, S, d, I["emask"], False, -1e9, 0.0)
                co = self.lin(ctx, p + ".co.w", p + ".co.b", H, H)
                y2 = self.ln(co, y1, p + ".ln2.w", p + ".ln2.b", H, 0.0, None, eps)
                a = self.lin(y2, p + ".fi.w", p + ".fi.b", dc.intermediate_size, H, gelu=True)
                fo = self.lin(a, p + ".fo.w", p + ".fo.b", H, dc.intermediate_size)
                y = self.ln(fo, y2, p + ".ln3.w", p + ".ln3.b", H, 0.0, None, eps)
            if pre is not None:
                # (the LM head keeps LayerNorm + Linear as two launches: 1908 workgroups of the LN-in kernel, each holding
                # gamma / beta in registers, stream the 47 MB of vocabulary weights at a quarter of the plain kernel's rate)
                y = self.ln(pre, None, lnp + ".w", lnp + ".b", H, 0.0, None, eps)
            return self.lin(y, "lm.w", "lm.b", Vp, H).t[:, :V]         # [Bn, V] view of the arena (row stride Vp), activation dtype

        return encode, one_token, st

    @staticmethod
    def _fused_sampling(P, vocab):
        """The fused sampling kernel covers the reference's settings (generate.py:138-141,177-180: top_k 7, top_p 0; BERT's
        30522-token vocabulary) and, since ABI 6, any top_k and top_p; only a vocabulary beyond one CU's LDS takes the torch-op
        form of the filters, issued eagerly step by step (no captured token graph)."""
        return vocab <= ops.SAMPLE_MAX_VOCAB        # (any top_k, any top_p: both filters run inside the sampling launch since ABI 6)

    @staticmethod
    def _sampling_step(logits, cur, pos, hist, P, u_row):
        """One step of models/visual_dialog_model.py:96-108 on static buffers: cur[pos] <- the token drawn from `logits`
        (temperature, n-gram ban against `hist`, top-k / top-p, softmax, inverse-CDF draw from the uniforms `u_row`).
        `cur` is the TIME-MAJOR id buffer [L0 + max_seq_len, B]: position t of all rows is one contiguous row, which the next
        token step's embedding reads as it is.  Free of host synchronisation and of generator state, so the token graph
        captures it together with the decoder stack."""
        from . import decoding
        if Engine._fused_sampling(P, logits.shape[-1]):
            # the n-gram ban (utils/decoding_utils.py:38-77) runs inside the sampling launch: `hist` and the time-major id buffer
            # are all it needs (round 4 built a [B, V + 1] mask with ten torch launches per token: +2 ms per questioner decode)
            ops.sample_topk(logits, P["temperature"], P["top_k"], u_row, cur[pos], None,
                            ngram=(hist, cur, pos, P["ngram"]) if P["ngram"] > 0 else None, top_p=P["top_p"])
            return
        last = logits.float() / P["temperature"]
        last = decoding.batch_ngram_blocking(last, hist, cur[:pos].t(), ngram_size=P["ngram"])
        last = decoding.batch_top_k_top_p_sampling(last, top_k=P["top_k"], top_p=P["top_p"])
        prob = torch.softmax(last, dim=-1)
        cur[pos] = decoding.draw_from_uniform(prob, u_row).view(-1)

    def _decode_session(self, ins, L0, max_seq_len, P):
        """hipGraph form of a decode call (generate.py's loop calls sample() with the same shapes and settings batch after
        batch): static copies of the inputs, one captured graph for `encode` and ONE for the whole token loop -- every decoder
        position's stack AND its sampling step (filters, softmax, draw, append), so that nothing of the loop is issued from
        the host at replay (round 1-2 replayed one graph per position and ran ~25 small torch kernels per step eagerly in
        between: the loop was bound by host issue).
        Returns (refresh(ins, uniforms), run_encode(), run_tokens(), st, cur, last_logits)."""
        static = tuple(x.clone() if x is not None else None for x in ins)
        ids, segs, dec_ids = static[3], static[4], static[6]
        Bn, dev = ids.shape[0], ids.device
        steps = L0 + max_seq_len - 1
        cur = torch.zeros(L0 + max_seq_len, Bn, dtype=torch.long, device=dev)      # time-major (see _sampling_step)
        u_buf = torch.zeros(max_seq_len, Bn, dtype=torch.float32, device=dev)
        encode, one_token, st = self._decode_plan(static, L0, max_seq_len)
        from .graph import capture, gc_quiet
        with gc_quiet():
            g_enc = torch.cuda.CUDAGraph()
            with capture(g_enc):
                encode()
                hist = ids * (segs == 0).long()
            g_dec = torch.cuda.CUDAGraph()
            with capture(g_dec, pool=g_enc.pool(), quiesce=False):
                cur[:L0] = dec_ids.t()
                for t in range(steps):
                    logits = one_token(cur[t], t)
                    if t >= L0 - 1:
                        self._sampling_step(logits, cur, t + 1, hist, P, u_buf[t - (L0 - 1)])

        def refresh(new, uniforms):
            for dst, src in zip(static, new):
                if dst is not None:
                    dst.copy_(src)
            u_buf.copy_(uniforms)

        return refresh, g_enc.replay, g_dec.replay, st, cur, logits

    @torch.no_grad()
    def sample(self, feats, loc, img_mask, ids, segs, att_mask, dec_ids, temperature=1.0, top_k=0, top_p=0.0,
               ngram_blocking_size=0, max_seq_len=18, uniforms=None, **_):
        """models/visual_dialog_model.py:74-120: 18 sampling steps (temperature, n-gram blocking, top-k / top-p, multinomial
        draw, [PAD] after the first [SEP]).  The reference re-runs the whole decoder on the growing prefix and re-projects
        the cross-attention K/V of all 37+T encoder states in all 12 layers at every step (use_cache=False); here the
        encoder, VLFusion and the cross K/V projection run once, and each step feeds ONE token per row through the stack,
        appending its self-attention K/V to a [B, Umax, H] cache per layer.  Same arithmetic, O(U) instead of O(U^2).
        From the second call with the same shapes and sampling settings on (params['amd_decode_graph'], default on) the
        device work is replayed from two captured hipGraphs (the encoder side; the whole token loop incl. its sampling
        steps): ~3000 launches per call leave the host.
        Token-id work (filters, n-gram ban, EOS fill) is integer-exact torch index plumbing (decoding.py), free of host syncs.
        Token t is drawn by inverse CDF from uniforms[t] ([max_seq_len, B] in (0, 1); drawn from torch's default generator when
        the caller passes none) instead of torch.multinomial (whose stream is device specific) -- the same rule the oracle
        applies to the reference, so sampled ids can be compared under real sampling."""
        from . import decoding
        dc = self.dec_cfg
        if segs is None:
            segs = torch.zeros_like(ids)
        ins = (feats, loc, img_mask, ids, segs, att_mask, dec_ids)
        L0 = dec_ids.shape[1]
        Bn = ids.shape[0]
        P = dict(temperature=float(temperature), top_k=int(top_k), top_p=float(top_p), ngram=int(ngram_blocking_size))
        sig = (L0, max_seq_len, tuple(sorted(P.items()))) + tuple((tuple(x.shape), x.dtype) if x is not None else None for x in ins)
        if uniforms is None:
            # the call's randomness, drawn ONCE from torch's default CUDA generator (eagerly: no generator state inside the
            # captured graphs); every step then draws by inverse CDF -- the same distribution as the reference's
            # torch.multinomial (whose stream is device specific anyway), and the same ids from eager issue and graph replay
            u = torch.rand(max_seq_len, Bn, device=ids.device, dtype=torch.float32).clamp_min_(1e-12)
        else:
            u = uniforms.to(ids.device, torch.float32)
            if u.dim() != 2 or u.shape[0] < max_seq_len or u.shape[1] != Bn:
                raise GstvdError("uniforms must be [max_seq_len = %d, batch = %d] (one draw per step and row), got %s"
                                 % (max_seq_len, Bn, tuple(u.shape)))
            u = u[:max_seq_len].contiguous()
        use_graph = bool(self.model.params.get("amd_decode_graph", True)) and self._fused_sampling(P, dc.vocab_size)
        # parameters edited since the last call (load_state_dict, an optimizer step): the captured graphs read the flat
        # buffers / bf16 shadow, so bring those up to date OUTSIDE the graphs; a re-materialised buffer drops the sessions
        self.prepare(ids.device)
        sess = self._decode_sessions.get(sig) if use_graph else None
        if sess is not None:
            refresh, run_encode, run_tokens, dst, cur, last_logits = sess
            refresh(ins, u)
            run_encode()
            run_tokens()
            cur, logits = cur.clone(), last_logits
        else:
            run_encode, one_token, dst = self._decode_plan(ins, L0, max_seq_len)
            run_encode()
            hist = ids * (segs == 0).long()
            cur = torch.zeros(L0 + max_seq_len, Bn, dtype=torch.long, device=ids.device)
            cur[:L0] = dec_ids.t()
            calls0 = _libmod.N_CALLS[0]
            for t in range(L0 + max_seq_len - 1):
                logits = one_token(cur[t], t)
                if t >= L0 - 1:                            # (earlier positions only consume the given prefix)
                    self._sampling_step(logits, cur, t + 1, hist, P, u[t - (L0 - 1)])
            self.decode_lib_calls_per_token = (_libmod.N_CALLS[0] - calls0) / float(L0 + max_seq_len - 1)
        self.last = dict(decode_logits=logits.float())    # last position's raw logits (tests / debugging)
        # the encoder side of this call (cross-attention K/V of all layers, masks) stays valid in the arena until the next
        # engine call: `rescore_sampled` scores the sampled answer against it without a second encoder pass
        out = decoding.pad_after_eos(cur[L0:].t().contiguous(), dc.eos_token_id, dc.pad_token_id)
        if use_graph and sess is None:
            # first call with these shapes ran eagerly (it also initialised every lazily built table / attribute / arena
            # chunk); capture now so the next batch replays
            if len(self._decode_sessions) >= 4:
                self._decode_sessions.clear()
            self._decode_sessions[sig] = self._decode_session(ins, L0, max_seq_len, P)
        # (after the capture: capturing runs the Python side of encode() again -- which rewinds the arena bookkeeping and drops
        # this marker -- but executes nothing, so the eager call's encoder states are still what the arena holds)
        self._last_decode = (dst, ids.shape[0], self.arena)
        return out


    @torch.no_grad()
    def rescore_sampled(self, dec_ids, dec_mask=None, loss_reduction=False):
        """The "ppl trick" of generate.py:183-211 fused onto the decode call that produced the answer: ONE teacher-forced
        decoder pass over `dec_ids` against the encoder states / cross-attention K/V that the last `sample()` call left in
        the arena (same context by construction: the answer was sampled from it) -- no second encoder run, no second K/V
        projection.  Same conventions as the reference's labels=None branch (visual_dialog_decoder.py:53-57): labels are
        the ids shifted left, `dec_ids` has [SEP] -> [PAD] in place.  Returns (loss, logits) like `step`."""
        ld = getattr(self, "_last_decode", None)
        if ld is None or ld[2] is not self.arena or ld[1] != dec_ids.shape[0]:
            raise GstvdError("rescore_sampled: no decode state of a matching sample() call to reuse")
        st = ld[0]
        dc = self.dec_cfg
        self.arena.rewind(st["mark"])
        self.tape, self.rec, self.tag, self.train = [], False, "t", False
        self.main = torch.cuda.current_stream()
        labels = dec_ids.new_zeros(dec_ids.shape)
        labels[:, :-1] = dec_ids[:, 1:].clone()
        dec_ids.masked_fill_(dec_ids == dc.eos_token_id, dc.pad_token_id)
        I = dict(st["I"])
        Bn, U, V = I["B"], dec_ids.shape[1], dc.vocab_size
        I["U"] = U
        I["dec_ids"] = dec_ids.contiguous().view(-1)
        I["dmask"] = dec_mask.float().contiguous() if dec_mask is not None else None
        _, logits = self.decoder(None, I, kv=st["kv"])
        Md = Bn * U
        row_loss, lse, stats = self.vec(Md), self.vec(Md), self.vec(4)
        ops.ce_fwd(logits.t, labels.contiguous().view(-1), Md, V, row_loss, lse, stats, ignore_index=dc.pad_token_id)
        self._last_decode = None                  # the decode scratch behind the mark has been overwritten
        loss = stats[2].clone() if loss_reduction else row_loss.clone()
        return loss, logits.t.view(Bn, U, self.flat.Vp)[:, :, :V].to(torch.float32, copy=True)


class _StepFn(torch.autograd.Function):
    """Bridges the hand-written backward into torch.autograd so `loss.backward()` (train_gen.py:324) and
    d loss / d enc_image_features (FGSM in evaluate_gen_attack.py:101-131) keep working."""

    @staticmethod
    def forward(ctx, anchor, feats, engine, st, loss_raw):
        ctx.engine, ctx.st = engine, st
        ctx.has_feats = feats is not None
        ctx.feats_shape = feats.shape if feats is not None else None
        return loss_raw.clone()

    @staticmethod
    def backward(ctx, gloss):
        dfe = ctx.engine.backward(ctx.st, gloss)
        if ctx.has_feats and dfe is not None:
            dfe = dfe.view(ctx.feats_shape)
        return None, dfe, None, None, None


# ---- stand-alone encoder / decoder calls (inference plumbing; no autograd) --------------------------------
def _owner_engine(module, kind):
    eng = getattr(module, "_standalone_engine", None)
    if eng is None:
        raise GstvdError("%s.forward outside an EncoderDecoderModel is not supported by the MI355X engine; "
                         "call EncoderDecoderModel(...) (the reference scripts only ever do that)" % kind)
    return eng


def standalone_encoder_forward(module, input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask):
    eng = _owner_engine(module, "VisualDialogEncoder")
    with torch.no_grad():
        eng._begin(input_ids.device, False)
        dummy = input_ids.new_zeros(input_ids.shape[0], 1)
        I = eng._inputs(image_feat, image_loc, image_attention_mask, input_ids, token_type_ids, attention_mask, dummy, None)
        xt, xv = eng.encoder(I)
        Bn = I["B"]
        return xt.t.view(Bn, I["T"], -1).float(), xv.t.view(Bn, I["R"], -1).float()


def standalone_decoder_forward(module, dec_ids, attention_mask, enc_hidden, enc_mask, labels, loss_reduction):
    """VisualDialogDecoder.forward on caller-supplied encoder states (models/visual_dialog_decoder.py:33-86), inference
    only: the training graph lives in EncoderDecoderModel's single autograd function.  Same conventions as the reference:
    labels=None -> labels are the ids shifted left and the caller's `decoder_input_ids` has [SEP] replaced by [PAD] in place."""
    eng = _owner_engine(module, "VisualDialogDecoder")
    if torch.is_grad_enabled() and (enc_hidden.requires_grad or module.training):
        raise GstvdError("VisualDialogDecoder.forward alone is inference only on the MI355X engine (eval() + torch.no_grad()); "
                         "train through EncoderDecoderModel(...)")
    with torch.no_grad():
        eng._begin(dec_ids.device, False)
        eng.train = False
        dc = eng.dec_cfg
        if labels is None:
            labels = dec_ids.new_zeros(dec_ids.shape)
            labels[:, :-1] = dec_ids[:, 1:].clone()
            dec_ids.masked_fill_(dec_ids == dc.eos_token_id, dc.pad_token_id)
        Bn, S, H = enc_hidden.shape
        U, V = dec_ids.shape[1], dc.vocab_size
        if H != dc.hidden_size:
            raise GstvdError("encoder_hidden_states width %d != decoder hidden size %d" % (H, dc.hidden_size))
        enc = eng.act(Bn * S, H)
        enc.t.copy_(enc_hidden.reshape(Bn * S, H))
        em = enc_mask if enc_mask is not None else torch.ones(Bn, S, device=dec_ids.device)
        dm = attention_mask if attention_mask is not None else torch.ones(Bn, U, device=dec_ids.device)
        I = dict(B=Bn, T=S, R=0, U=U, emask=em.float().contiguous(), dmask=dm.float().contiguous(),
                 dec_ids=dec_ids.contiguous().view(-1))
        y, logits = eng.decoder(enc, I)
        Md = Bn * U
        row_loss, lse, stats = eng.vec(Md), eng.vec(Md), eng.vec(4)
        ops.ce_fwd(logits.t, labels.contiguous().view(-1), Md, V, row_loss, lse, stats, ignore_index=dc.pad_token_id)
        loss = stats[2].clone() if loss_reduction else row_loss.clone()
        return loss, logits.t.view(Bn, U, eng.flat.Vp)[:, :, :V].float()
