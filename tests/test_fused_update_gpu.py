"""Single-GPU training applies AdamW in the epilogue of the grouped weight-gradient launch (gstvd_gemm_grouped_adamw + the
gstvd_adamw_blocks remainder pass; train_gen.py:324-329 loss.backward(); optimizer.step(); optimizer.zero_grad()).  The bar:
BIT-IDENTICAL parameters, moments and bf16 shadow weights to the two launches it replaces (gstvd_gemm_grouped, gstvd_adamw) --
at op level on ragged shapes, and on the model over several steps, eager and under hipGraph replay."""
import os
import sys

import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sc():
    from gst_visdial_amd import selfcheck
    return selfcheck


class _Flat(object):
    """A stand-in for the optimizer's view of the flat buffers: tensors at 64-element aligned offsets, one (lr, wd) segment per
    tensor and per gap, exactly the layout optim.FusedAdamW._build produces."""

    def __init__(self, shapes, gap=192, seed=0):
        from gst_visdial_amd import _lib as L
        g = torch.Generator().manual_seed(seed)
        self.offs, ends, hp, off = [], [], [], 1024
        ends.append(off); hp += [1e-3, 0.0]                       # something in front of the first weight (an embedding, say)
        self.seg_of = {}
        for i, (M, N) in enumerate(shapes):
            self.offs.append(off)
            self.seg_of[off] = len(ends)
            off += M * N
            ends.append(off); hp += [1e-3 * (1 + i % 3), 0.01 if i % 2 else 0.0]
            off += gap                                             # a bias / LayerNorm-sized tensor between the weights
            ends.append(off); hp += [2e-3, 0.0]
        self.n = off
        self.P0 = (torch.randn(self.n, generator=g) * 0.05).to(DEV)
        self.M0 = (torch.randn(self.n, generator=g) * 0.01).to(DEV)
        self.V0 = (torch.rand(self.n, generator=g) * 1e-4 + 1e-6).to(DEV)
        self.G0 = (torch.randn(self.n, generator=g) * 0.02).to(DEV)   # the gaps' gradients; the weights' slots get overwritten
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=DEV)
        self.seg_ends_host = ends
        self.hp = torch.tensor(hp, dtype=torch.float32, device=DEV)
        self.step = torch.full((1,), 3.0, device=DEV)
        self.L = L

    def state(self):
        return self.P0.clone(), self.M0.clone(), self.V0.clone(), torch.zeros(self.n, dtype=torch.bfloat16, device=DEV), self.G0.clone()


class _Fuse(object):
    def __init__(self, fl, P, M, V, S, G, write_grad):
        d = fl.L.AdamFuse()
        d.grad_base, d.param, d.m, d.v, d.shadow_bf16 = G.data_ptr(), P.data_ptr(), M.data_ptr(), V.data_ptr(), S.data_ptr()
        d.step, d.beta1, d.beta2, d.eps, d.grad_scale, d.write_grad = fl.step.data_ptr(), 0.9, 0.999, 1e-6, 0.5, int(write_grad)
        self.d, self.fl, self.g0 = d, fl, G.data_ptr()

    def desc(self):
        return self.d

    def cover(self, c, M, N, ldc):
        off = (c - self.g0) // 4
        seg = self.fl.seg_of.get(off)
        return (0, ()) if seg is None else (self.fl.hp.data_ptr() + 8 * seg, (off,))


# (out features M, in features N, batch rows K): full tiles, ragged rows / columns / K, a single-tile problem, a 2-round one
SHAPES = [(768, 768, 4096), (300, 64, 37), (1024, 1024, 592), (256, 256, 400), (520, 196, 100), (3072, 768, 1000), (64, 3072, 256)]


@pytest.mark.parametrize("write_grad", [False, True])
def test_grouped_adamw_is_bit_identical_to_grouped_gemm_then_adamw(write_grad):
    from gst_visdial_amd import ops
    fl = _Flat([(M, N) for M, N, _ in SHAPES], seed=5)
    g = torch.Generator().manual_seed(9)
    dys = [(torch.randn(K, M, generator=g) * 0.1).to(DEV).bfloat16() for M, N, K in SHAPES]
    xs = [torch.randn(K, N, generator=g).to(DEV).bfloat16() for M, N, K in SHAPES]
    accumulate = [False, False, True, False, False, False, False]      # problem 2 adds to an existing gradient: never fused

    def run(fused):
        P, Mo, V, S, G = fl.state()
        grp = ops.GemmGroup(torch.device(DEV), a_km=True, b_km=True)
        biases = [torch.zeros(M, device=DEV) for M, N, K in SHAPES]
        for i, ((M, N, K), dy, x, off, acc) in enumerate(zip(SHAPES, dys, xs, fl.offs, accumulate)):
            # every other problem also asks for its bias gradient (column sums of dY) out of the same launch, as the engine does
            cs = biases[i] if (i % 2 == 0 and grp.colsum_capable(dy)) else None
            grp.add(dy, x, G[off:off + M * N].view(M, N), M, N, K, acc, colsum_out=cs)
        if fused:
            done = grp.flush(fuse=_Fuse(fl, P, Mo, V, S, G, write_grad))
            assert sorted(done) == sorted(o for o, a in zip(fl.offs, accumulate) if not a)
            skip = torch.zeros(len(fl.seg_ends_host), dtype=torch.uint8)
            for o in done:
                skip[fl.seg_of[o]] = 1
            need, start = set(), 0
            for i, end in enumerate(fl.seg_ends_host):
                if not skip[i]:
                    need.update(range(start // 1024, (end - 1) // 1024 + 1))
                start = end
            blocks = torch.tensor(sorted(need), dtype=torch.int32, device=DEV)
            ops.adamw_blocks(P, G, Mo, V, S, fl.seg_end, fl.hp, fl.step, blocks, skip.to(DEV), grad_scale=0.5, begin=0, end=fl.n)
        else:
            assert grp.flush() == ()
            ops.adamw(P, G, Mo, V, S, fl.seg_end, fl.hp, fl.step, grad_scale=0.5, begin=0, end=fl.n)
        torch.cuda.synchronize()
        return P, Mo, V, S, G, torch.cat(biases)

    ref, got = run(False), run(True)
    for name, a, b in zip(("param", "m", "v", "shadow"), ref[:4], got[:4]):
        assert torch.equal(a, b), (name, (a.float() - b.float()).abs().max().item(), int((a != b).sum()))
    assert torch.equal(ref[5], got[5]) and ref[5].abs().sum().item() > 0      # the bias gradients of the same launch
    assert not torch.equal(ref[0], fl.P0)                        # and something happened
    if write_grad:                                               # the launch also left dW where `.grad` looks for it
        assert torch.equal(ref[4], got[4])


def test_adamw_blocks_honours_range_and_segment_skip():
    """The remainder pass alone: blocks that straddle the range's start / end and a skipped segment stay untouched outside
    [begin, end) resp. inside the segment; everything else equals the full pass."""
    from gst_visdial_amd import ops
    fl = _Flat([(96, 64), (128, 32), (64, 64)], gap=320, seed=2)
    P, Mo, V, S, G = fl.state()
    ops.adamw(P, G, Mo, V, S, fl.seg_end, fl.hp, fl.step, begin=0, end=fl.n)
    P2, M2, V2, S2, G2 = fl.state()
    lo, hi = fl.offs[0] + 64 * 8, fl.offs[2] + 64 * 10            # 64-element aligned, inside tensors, not on 1024 boundaries
    skip = torch.zeros(len(fl.seg_ends_host), dtype=torch.uint8)
    skip[fl.seg_of[fl.offs[1]]] = 1
    blocks = torch.arange(lo // 1024, (hi - 1) // 1024 + 1, dtype=torch.int32, device=DEV)
    ops.adamw_blocks(P2, G2, M2, V2, S2, fl.seg_end, fl.hp, fl.step, blocks, skip.to(DEV), begin=lo, end=hi)
    torch.cuda.synchronize()
    inside = torch.zeros(fl.n, dtype=torch.bool, device=DEV)
    inside[lo:hi] = True
    inside[fl.offs[1]:fl.offs[1] + 128 * 32] = False
    assert torch.equal(P2[inside], P[inside]) and torch.equal(M2[inside], Mo[inside]) and torch.equal(V2[inside], V[inside])
    assert torch.equal(S2[inside], S[inside])
    assert torch.equal(P2[~inside], fl.P0[~inside]) and torch.equal(M2[~inside], fl.M0[~inside]) and torch.equal(V2[~inside], fl.V0[~inside])


def _atomic_fed(model):
    """Flat ranges of the embedding tables and of the image-location projection: their gradients are accumulated with fp32 atomic
    adds (gstvd_ln_bwd in EMBED mode, gstvd_locgrad), whose order -- and so whose last bit -- differs from run to run of the SAME
    code (two runs of the two-launch path differ there too); everything else in a step is deterministic."""
    m = torch.zeros(model.engine.flat.P.numel(), dtype=torch.bool, device=DEV)
    names = {id(p): n for n, p in model.named_parameters()}
    for p, off in model.engine.flat.items:
        n = names.get(id(p), "")
        if any(t in n for t in ("word_embeddings", "position_embeddings", "token_type_embeddings", "image_location_embeddings")):
            m[off:off + p.numel()] = True
    return m


def _run_model(fuse, chunk, keep, steps, graph):
    s = sc()
    from gst_visdial_amd import ops
    from gst_visdial_amd.graph import GraphedStep
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    g = load_npz("tiny_train.npz")
    model, params, cfg = s.build_tiny_model("bf16", DEV, seed=21, cfg_file="tiny_cfg_dropout.json")
    model.train()
    kw = s.golden_batch(g, DEV)
    opt = FusedAdamW(model, lr=1e-3, warmup_steps=2, t_total=10)
    pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=chunk, keep_grads=keep)
    pipe.fuse_update = fuse

    def step():
        loss, _ = model(**kw)
        loss.backward()
        opt.step()
        opt.scheduler_step()
        opt.zero_grad()
        return loss

    names = []
    if graph:
        replay = GraphedStep(step, warmup=2)
        for _ in range(steps - 2):
            opt.upload_lr()
            replay()
    else:
        prof = ops.Profiler()
        with prof:
            for _ in range(steps):
                step()
        names = [r[0] for r in prof.records]
    torch.cuda.synchronize()
    fl = model.engine.flat
    pn = {id(p): n for n, p in model.named_parameters()}
    return dict(param=fl.P.clone(), m=opt.m.clone(), v=opt.v.clone(), shadow=fl.S.clone(), grad=fl.G.clone(), names=names,
                slices=len(pipe.slices), atomic=_atomic_fed(model), items=[(off, p.numel(), pn.get(id(p), "?")) for p, off in fl.items])


@pytest.mark.isolated
@pytest.mark.parametrize("chunk", [1 << 40, 60000], ids=["one_slice", "slices"])
def test_model_step_with_the_update_in_the_weight_gradient_launch_is_bit_identical(chunk):
    """bf16 tiny model, dropout on, one training step from the same initial state with the backward pipeline's fusion
    (GSTVD_FUSE_UPDATE, pipe.fuse_update) off and on: parameters, moments and bf16 shadow weights bit-identical wherever a step is
    deterministic at all (everything but the atomically accumulated embedding tables: 1e-6 there); the fused launch really ran;
    with keep_grads the flat gradient buffer is complete as well."""
    ref = _run_model(False, chunk, False, 1, False)
    got = _run_model(True, chunk, False, 1, False)
    det = ~ref["atomic"]
    assert det.sum() > 0.5 * det.numel()
    for k in ("param", "m", "v", "shadow"):
        a, b = ref[k], got[k]
        assert torch.equal(a[det], b[det]), (k, (a.float() - b.float())[det].abs().max().item(), int((a != b)[det].sum()))
        # the atomically accumulated tables (embedding gradients: the order of the atomic adds is the hardware's) differ in the last
        # fp32 bits from run to run; that can flip the rounding of a bf16 shadow weight -- one ulp (2^-7 relative) there, 1e-6 elsewhere
        d = (a.float() - b.float()).abs()
        if k == "shadow" and a.dtype == torch.bfloat16:
            assert bool((d <= a.float().abs() * 2.0 ** -7 + 1e-6).all()), k
        else:
            assert d.max().item() < 1e-6, k
    assert not any("grouped_adamw" in n for n in ref["names"]) and any("grouped_adamw" in n for n in got["names"]), got["names"][:20]
    assert (chunk > 1 << 30) == (got["slices"] == 1)
    kept = _run_model(True, chunk, True, 1, False)
    for k in ("param", "m", "v", "shadow", "grad"):
        assert torch.equal(ref[k][det], kept[k][det]), k


@pytest.mark.isolated
def test_fused_update_under_hipgraph_replay_follows_the_schedule():
    """Two eager warm-up steps + three replays with a moving learning rate (the launch reads each weight's lr from the device
    table at run time): the captured fused launch tracks the two-launch trajectory (not bitwise over five steps -- see _atomic_fed)."""
    ref = _run_model(False, 1 << 40, False, 5, True)
    got = _run_model(True, 1 << 40, False, 5, True)
    for k in ("param", "m", "v"):
        a, b = ref[k], got[k]
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item()), (k, (a - b).abs().max().item())
    moved = (ref["param"] - _run_model(False, 1 << 40, False, 1, False)["param"]).abs().max().item()
    assert moved > 1e-4                                  # the replays really trained


@pytest.mark.isolated
def test_full_size_step_with_the_fused_update_is_bit_identical_to_the_two_launches():
    """The 388 M-parameter model at the bench shape (16 rows, seq_len 256, bf16, dropout on), one training step from the same
    state through the single-slice pipeline with and without the fold: 5309 tiles of 197 problems, the fused Q|K|V blocks (three
    tensors, one problem), the LM head with its rows padded to a multiple of 64, K = 400 / 592 / 4096.  Bit-identical parameters,
    moments and shadow weights outside the atomically accumulated tables; > 90 % of the parameters took the fused path."""
    import bench
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline

    def run(fuse):
        model, params = bench.build_model(torch.device(DEV), "bf16", seed=3)
        model.train()
        V = model.decoder.config.vocab_size
        batch = bench.synthetic_rows(16, 256, 37, 25, 2048, V, 99, DEV)
        opt = FusedAdamW(model, lr=1e-3, warmup_steps=0, t_total=100)
        pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=1 << 40)
        pipe.fuse_update = fuse
        loss, _ = model(**batch)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        fl = model.engine.flat
        nfused = 0
        if fuse:
            (key, (blocks, skip)), = opt._remainders.items()
            seg_len = torch.tensor([e - s for s, e in zip([0] + opt.seg_ends_host[:-1], opt.seg_ends_host)])
            nfused = int(seg_len[skip.cpu().bool()].sum())
        out = dict(param=fl.P.clone(), m=opt.m.clone(), v=opt.v.clone(), shadow=fl.S.clone(), atomic=_atomic_fed(model), loss=loss.item(),
                   nfused=nfused, n=fl.n_live)
        del model, opt, pipe
        torch.cuda.empty_cache()
        return out

    ref, got = run(False), run(True)
    assert ref["loss"] == got["loss"]
    det = ~ref["atomic"]
    for k in ("param", "m", "v", "shadow"):
        a, b = ref[k], got[k]
        assert torch.equal(a[det], b[det]), (k, (a.float() - b.float())[det].abs().max().item(), int((a != b)[det].sum()))
        # (atomically accumulated tables: the last fp32 bits follow the order of the atomic adds, which can flip the rounding of a bf16
        # shadow weight -- one ulp there, see the model-level test above)
        d = (a.float() - b.float()).abs()
        if k == "shadow" and a.dtype == torch.bfloat16:
            assert bool((d <= a.float().abs() * 2.0 ** -7 + 1e-6).all()), k
        else:
            assert d.max().item() < 1e-6, k
    assert got["nfused"] > 0.9 * got["n"], (got["nfused"], got["n"])
    assert not torch.equal(ref["param"], ref["shadow"].float())        # (sanity: fp32 masters differ from their bf16 shadows)
