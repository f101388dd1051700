"""Round 5 on the GPU: the XCD placement of the grouped weight-gradient launch (results must not depend on it), the tied
decoder embedding (a second writer into a weight's gradient slot), `.grad` of weights the launch updated itself,
pipe.skip_update_once() (train_gen.py:326-329, iteration 0) and AdamW's fast square root / reciprocal against the IEEE sequence."""
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


# ---- placement --------------------------------------------------------------------------------------------------------------
SHAPES = [(768, 768, 400), (3072, 768, 1000), (300, 64, 37), (1024, 1024, 592), (768, 3072, 400), (2304, 768, 1024), (520, 196, 100),
          (768, 768, 1000), (3072, 768, 400), (256, 256, 400), (1024, 2048, 592), (64, 3072, 256), (768, 1024, 1000)]


@pytest.mark.parametrize("mode", [3])
def test_grouped_launch_results_do_not_depend_on_the_block_map(mode, monkeypatch):
    """gstvd_gemm_grouped with block_map_dev: every tile of every problem runs exactly once whatever the order -- dW and the bias
    column sums are bit-identical to the library's own order (same per-tile arithmetic), incl. ragged shapes and idle blocks."""
    from gst_visdial_amd import ops
    g = torch.Generator().manual_seed(3)
    dys = [(torch.randn(K, M, generator=g) * 0.1).to(DEV).bfloat16() for M, N, K in SHAPES]
    xs = [torch.randn(K, N, generator=g).to(DEV).bfloat16() for M, N, K in SHAPES]

    def run(order):
        monkeypatch.setattr(ops, "GROUP_ORDER", order)
        monkeypatch.setattr(ops, "GROUP_ORDER_MIN_TILES", 0)
        grp = ops.GemmGroup(torch.device(DEV), a_km=True, b_km=True)
        outs = [torch.full((M, N), 7.0, device=DEV) for M, N, K in SHAPES]
        biases = [torch.zeros(M, device=DEV) for M, N, K in SHAPES]
        for i, ((M, N, K), dy, x) in enumerate(zip(SHAPES, dys, xs)):
            grp.add(dy, x, outs[i], M, N, K, i % 4 == 3, colsum_out=biases[i] if grp.colsum_capable(dy) else None)
        grp.flush()
        torch.cuda.synchronize()
        hit = list(grp.cache.values())[0]
        return outs, biases, hit[6]

    ref_o, ref_b, bm0 = run(0)
    got_o, got_b, bm = run(mode)
    assert bm0 is None and bm is not None and int((bm >= 0).sum()) == sum(((M + 255) // 256) * ((N + 255) // 256) for M, N, K in SHAPES)
    assert int((bm < 0).sum()) > 0                              # the queues of this table are ragged: idle blocks are exercised
    for (M, N, K), a, b, dy, x in zip(SHAPES, ref_o, got_o, dys, xs):
        assert torch.equal(a, b), (M, N, K)
    for a, b in zip(ref_b, got_b):
        assert torch.equal(a, b)
    want = dys[1].float().t() @ xs[1].float()
    assert (got_o[1] - want).abs().max().item() < 2e-2 * want.abs().max().item()


def test_bad_block_map_is_rejected():
    from gst_visdial_amd import _lib as L
    lib = L.load()
    z = torch.zeros(64, dtype=torch.int32, device=DEV)
    tab = torch.zeros(512, dtype=torch.uint8, device=DEV)
    rc = lib.gstvd_gemm_grouped(tab.data_ptr(), z.data_ptr(), 1, 9, L.BF16, L.F32, 1, 1, z.data_ptr(), 8, None)
    assert rc == -2                                             # GSTVD_E_SHAPE: a map shorter than the tile count


# ---- the tied LM head: two writers into one gradient slot ----------------------------------------------------------------------
def _untied_vs_tied_models(precision):
    """The golden tiny model WITHOUT train_gen.py:293: the decoder keeps its own embedding module, whose word table IS the LM
    head's weight (visual_dialog_decoder.py:329-335)."""
    import json
    import tempfile
    from gst_visdial_amd.modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "tiny_cfg.json")))
    d = tempfile.mkdtemp(prefix="gstvd_cfg_")
    json.dump(cfg["enc"], open(d + "/e.json", "w"))
    json.dump(cfg["dec"], open(d + "/d.json", "w"))
    params = dict(model_enc_config=d + "/e.json", model_dec_config=d + "/d.json", gpu_ids=[0], model="enc_dec_a", mode="vd_train",
                  batch_size=3, device=torch.device(DEV), amd_precision=precision, amd_seed=0)
    enc, dec = VisualDialogEncoder(params), VisualDialogDecoder(params)
    model = EncoderDecoderModel(params, enc, dec)
    sd = load_npz("tiny_state.npz")
    g = torch.Generator().manual_seed(11)
    # the fixture was written with shared embeddings: give the decoder's own module values of its own, tie the LM head to its table
    for k in list(sd):
        if k.startswith("decoder.decoder.bert.embeddings.") and "LayerNorm" not in k:
            sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
    sd["decoder.decoder.lm_head.decoder.weight"] = sd["decoder.decoder.bert.embeddings.word_embeddings.weight"]
    model.load_state_dict(sd, strict=True)
    return model.to(DEV), sd, cfg


@pytest.mark.parametrize("with_pipeline", [False, True], ids=["plain_backward", "pipeline_fused_update"])
def test_tied_lm_head_gradient_has_both_contributions(with_pipeline):
    """with the LM head tied to the decoder's own word embedding the table's gradient slot has TWO writers -- the
    LM-head weight-gradient GEMM (queued, launched later) and the embedding scatter-add.  Both must arrive, and a fused update
    must not be applied to that weight from the GEMM's accumulators alone."""
    from oracle import vd_oracle as O
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    model, sd, cfg = _untied_vs_tied_models("fp32")
    model.eval()
    g = load_npz("tiny_train.npz")
    eng = model.engine
    if with_pipeline:
        opt = FusedAdamW(model, lr=1e-3)
        pipe = BackwardPipeline(eng, optimizer=opt, chunk_elems=1 << 40, keep_grads=True)
    loss, logits = model(**sc().golden_batch(g, DEV))
    p_before = None
    if with_pipeline:
        eng.prepare(torch.device(DEV))
        p_before = eng.flat.P.clone()
    loss.backward()
    torch.cuda.synchronize()
    # oracle with the tie restored by hand: one tensor behind both keys
    osd = {k: v.clone() for k, v in sd.items()}
    wkey, lkey = "decoder.decoder.bert.embeddings.word_embeddings.weight", "decoder.decoder.lm_head.decoder.weight"
    tied = osd[wkey].clone().requires_grad_(True)
    osd[wkey] = osd[lkey] = tied
    osd["decoder.decoder.lm_head.decoder.bias"] = osd["decoder.decoder.lm_head.bias"]
    b = {k[4:]: v.clone() for k, v in g.items() if k.startswith("in::")}
    ref = O.model_forward(osd, cfg["enc"], cfg["dec"], b)
    ref["loss"].backward()
    assert abs(loss.item() - ref["loss"].item()) < 1e-5
    w = model.decoder.decoder.lm_head.decoder.weight
    assert w is model.decoder.decoder.bert.embeddings.word_embeddings.weight
    got = w.grad.float().cpu()
    want = tied.grad
    lm_only = want.clone()
    err = (got - want).abs().max().item()
    assert err < 2e-4 * want.abs().max().item() + 1e-7, err
    # the embedding rows of the decoder's input ids carry the scatter-add part: it must be there (a lost contribution shows here)
    ids = b["dec_input_ids"].reshape(-1).unique()
    assert (got[ids] - want[ids]).abs().max().item() < 2e-4 * want.abs().max().item() + 1e-7
    if with_pipeline:
        off = eng.flat.slots["lm.w"][0]
        n = want.numel()
        assert eng.flat.slots["lm.w"] == eng.flat.slots["demb.word"]
        # the update that WAS applied used the full gradient: compare with a plain AdamW step on the oracle gradient
        m = 0.1 * want
        v = 0.001 * want * want
        upd = p_before[off:off + n].cpu().view_as(want) - 1e-3 * ((1 - 0.999) ** 0.5 / (1 - 0.9)) * m / (v.sqrt() + 1e-6)
        upd = upd * (1 - 1e-3 * 0.01)
        now = eng.flat.P[off:off + n].cpu().view_as(want)
        assert (now - upd).abs().max().item() < 5e-6


# ---- .grad of weights the launch updated itself ------------------------------------------------------------------------------
def test_fused_weights_have_no_stale_grad_and_keep_grads_materialises_them():
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    g = load_npz("tiny_train.npz")
    for keep in (False, True):
        model, params, cfg = sc().build_tiny_model("bf16", DEV, seed=2)
        model.train()
        opt = FusedAdamW(model, lr=1e-3)
        pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=1 << 40, keep_grads=keep)
        for _ in range(2):
            loss, _ = model(**sc().golden_batch(g, DEV))
            loss.backward()
            opt.step()
            w = model.encoder.bert_pretrained.bert.encoder.layer[0].intermediate.dense.weight
            b = model.encoder.bert_pretrained.bert.encoder.layer[0].intermediate.dense.bias
            assert pipe.fuse_handle() is not None
            assert b.grad is not None and torch.isfinite(b.grad).all()
            if keep:
                assert w.grad is not None and w.grad.abs().sum().item() > 0
            else:
                assert w.grad is None                    # updated in the launch's epilogue, dW never stored: nothing stale to read
            opt.zero_grad()


# ---- iteration 0 of train_gen.py ------------------------------------------------------------------------------------------------
def test_skip_update_once_reproduces_the_reference_loops_first_iterations():
    """train_gen.py:324-329: `if iter_id > 0: optimizer.step(); optimizer.zero_grad()` -- iteration 0's gradients are neither applied
    nor zeroed and add to iteration 1's.  BackwardPipeline applies its update inside backward(); pipe.skip_update_once() makes
    the pipelined loop equal the plain FusedAdamW loop (which reproduces the reference's own 6-iteration run, test_round2_gpu)."""
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    g = load_npz("tiny_train.npz")

    def run(pipelined):
        model, params, cfg = sc().build_tiny_model("fp32", DEV, seed=1)
        model.eval()                                         # (dropout off: the two loops must agree to rounding)
        opt = FusedAdamW(model, lr=2e-3, warmup_steps=2, t_total=20)
        pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=60000) if pipelined else None
        losses = []
        for it in range(4):
            if pipelined and it == 0:
                pipe.skip_update_once()
            loss, _ = model(**sc().golden_batch(g, DEV))
            loss.backward()
            if it > 0:
                opt.step()
                opt.zero_grad()
            opt.scheduler_step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, model.engine.flat.P.detach().cpu().clone(), opt._sync_step()

    l0, p0, s0 = run(False)
    l1, p1, s1 = run(True)
    assert s0 == s1 == 3
    assert l0[0] == l0[1] and l1[0] == l1[1]                 # nothing moved at iteration 0
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5
    assert (p0 - p1).abs().max().item() < 5e-6


# ---- AdamW arithmetic ---------------------------------------------------------------------------------------------------------
def test_adamw_fast_sqrt_rcp_against_ieee_incl_tiny_second_moments():
    """adamw_update4 uses the hardware's ~1-ulp v_sqrt_f32 / v_rcp_f32 (shared with the weight-gradient launch's
    epilogue, with which it must agree bit for bit) instead of the IEEE sqrt / divide of pytorch_transformers' AdamW
    (exp_avg / (exp_avg_sq.sqrt() + eps)).  The deviation stays at the last bits of the UPDATE TERM -- which lr then scales to
    ~1e-9 of a weight -- also where the second moment is tiny or subnormal (v_sqrt_f32 does not take denormals)."""
    from gst_visdial_amd import ops
    n = 1 << 16
    g = torch.Generator().manual_seed(1)
    P = torch.randn(n, generator=g)
    P[::2] = 0.0                                                                               # (there the new weight IS the update term)
    grad = torch.randn(n, generator=g) * 10.0 ** (torch.rand(n, generator=g) * 30 - 28)       # gradients from 1e-28 to 100
    M = torch.zeros(n); V = torch.zeros(n)
    V[: n // 4] = 10.0 ** (torch.rand(n // 4, generator=g) * 20 - 44)                          # old second moments down to subnormal
    lr, wd, b1, b2, eps, t = 1e-3, 0.01, 0.9, 0.999, 1e-6, 5.0
    seg_end = torch.tensor([n], dtype=torch.int64, device=DEV)
    hp = torch.tensor([lr, wd], device=DEV)
    step = torch.full((1,), t, device=DEV)
    Pd, Gd, Md, Vd = P.to(DEV), grad.to(DEV), M.to(DEV), V.to(DEV)
    Sd = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    ops.adamw(Pd, Gd, Md, Vd, Sd, seg_end, hp, step, b1, b2, eps, 1.0)
    torch.cuda.synchronize()
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))                                 # the kernel's constants are floats
    b1f, b2f, lrf, wdf, epsf = f32(b1), f32(b2), f32(lr), f32(wd), f32(eps)
    c1, c2 = f32(1.0 - b1f), f32(1.0 - b2f)
    m = M.double() * b1f + grad.double() * c1
    v = V.double() * b2f + grad.double() ** 2 * c2
    bc = (1 - b2f ** t) ** 0.5 / (1 - b1f ** t)
    term = m / (v.sqrt() + epsf)
    p = (P.double() - lrf * bc * term) * (1 - lrf * wdf)
    assert torch.isfinite(Pd).all() and torch.isfinite(Vd).all()
    assert (Md.cpu().double() - m).abs().max().item() <= 1e-6 * m.abs().max().item()
    assert ((Vd.cpu().double() - v).abs() <= 1e-6 * v + 1.2e-38).all()        # (subnormal second moments may be flushed: far below eps^2)
    got = Pd.cpu().double()
    assert (got - p).abs().max().item() < 6e-7                                                 # one fp32 ulp of a weight of magnitude 4
    z = got[::2], p[::2]                                                                      # zero weights: p' = -lr bc term (1 - lr wd)
    rel = (z[0] - z[1]).abs() / (z[1].abs() + 1e-30)
    assert rel[z[1].abs() > 1e-20].max().item() < 5e-6                                         # the fast sqrt / rcp (and powf in bc) cost the TERM a few ulp, no more


# ---- the n-gram filter inside the sampling launch ---------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 4])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sampling_kernel_ngram_ban_equals_the_reference_filter(n, dtype):
    """utils/decoding_utils.py:38-77 (batch_ngram_blocking / _get_generated_ngrams) inside gstvd_sample_topk: the drawn ids equal
    those under the mask the restated reference loop builds (and under the torch-op mask of round 4), for histories with repeated
    n-grams, n-grams through special tokens (ignored), prefixes too short to match, and top_k = 1 (the survivor of the ban)."""
    from gst_visdial_amd import ops, decoding
    g = torch.Generator().manual_seed(40 + n)
    Bn, T, V, L = 6, 40, 700, 12
    hist = torch.randint(104, 140, (Bn, T), generator=g)                 # a small alphabet: many repeated n-grams
    hist[:, 0] = 101
    hist[0, 10] = 102; hist[1, 5:8] = 0; hist[2, -6:] = 0                # special tokens inside some windows
    for pos in (0, 1, 3, 7, L - 1):                                      # number of ids generated so far
        cur = torch.zeros(L, Bn, dtype=torch.long)
        for b in range(Bn):                                              # prefixes copied out of the history: the ban must bite
            s0 = int(torch.randint(0, T - 8, (1,), generator=g))
            cur[:pos, b] = hist[b, s0:s0 + pos] if pos > 0 else cur[:pos, b]
        logits = torch.randn(Bn, V, generator=g)
        # make the tokens that WOULD be banned the most likely ones, so that a missed ban changes the draw
        ref_masked = decoding._ngram_blocking_loop(torch.zeros(Bn, V), hist, cur[:pos].t(), ngram_size=n)
        logits = logits + 8.0 * (ref_masked == -float("inf")).float()
        lg = logits.to(DEV).to(dtype)
        u = torch.rand(Bn, generator=g).clamp(1e-3, 1 - 1e-3).to(DEV)
        hist_d, cur_d = hist.to(DEV), cur.to(DEV)
        for top_k in (1, 7):
            out_k = torch.zeros(Bn, dtype=torch.long, device=DEV)
            ops.sample_topk(lg, 0.7, top_k, u, out_k, None, ngram=(hist_d, cur_d, pos, n))
            want_mask = decoding._ngram_blocking_loop(torch.zeros(Bn, V), hist, cur[:pos].t(), ngram_size=n) == -float("inf")
            out_m = torch.zeros(Bn, dtype=torch.long, device=DEV)
            ops.sample_topk(lg, 0.7, top_k, u, out_m, want_mask.to(DEV))
            assert torch.equal(out_k, out_m), (n, pos, top_k, out_k.tolist(), out_m.tolist())
            tm = decoding.ngram_banned_mask(hist_d, cur_d[:pos].t(), n, V, DEV)
            if tm is not None:
                assert torch.equal(tm[:, :V].cpu(), want_mask)
            assert not want_mask[torch.arange(Bn), out_k.cpu()].any()     # no banned token is ever drawn
        if pos >= n - 1 and n > 1:
            assert want_mask.any()                                        # (the cases really ban something)


# ---- dropout keep bits from forward to the one-pass backward --------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 3, 256, 256), (1, 2, 100, 200), (2, 2, 64, 65), (1, 1, 330, 256)])
def test_attention_keep_bits_equal_the_hashed_mask_and_the_backward_that_reads_them(shape):
    """gstvd_attn_t.drop_bits: forward writes the keep bit of every probability it drops / keeps (16 x 16 tiles, four ballots each);
    the one-pass backward reads them instead of hashing each draw again.  The bits ARE the counter-hash mask (checked against the
    library's mask probe), so the backward's outputs are bit-identical with and without them; ragged query / key counts included."""
    from gst_visdial_amd import ops
    Bn, nh, Lq, Lk = shape
    d, p = 64, 0.1
    H = nh * d
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(Bn * max(Lq, Lk), 3 * H, generator=g) * 0.5).to(DEV).bfloat16()
    Q, K, V = qkv[:Bn * Lq, :H], qkv[:Bn * Lk, H:2 * H], qkv[:Bn * Lk, 2 * H:]
    km = torch.ones(Bn, Lk, device=DEV)
    km[0, Lk - Lk // 4:] = 0
    rng = ops.Rng(DEV, seed=9)
    n = ops.attn_keep_bits_shape(Bn, nh, Lq, Lk, d, torch.bfloat16, False, p)
    assert n == Bn * nh * ((Lq + 15) // 16) * ((Lk + 15) // 16) * 4
    bits = torch.zeros(n, dtype=torch.int64, device=DEV)
    outs = []
    for use_bits in (True, False):
        O = torch.empty(Bn * Lq, H, device=DEV, dtype=torch.bfloat16)
        lse = torch.empty(Bn, nh, Lq, device=DEV)
        a = ops.attn_desc(Q, K, V, O, lse, km, Bn, nh, Lq, Lk, d, mask_neg=-10000.0, drop_p=p, site=33, rng=rng,
                          drop_bits=bits if use_bits else None)
        ops.attn_fwd(a)
        dO = (torch.randn(Bn * Lq, H, generator=torch.Generator().manual_seed(6)) * 0.3).to(DEV).bfloat16()
        dQ, dK, dV = torch.full_like(Q, float("nan")), torch.full_like(K, float("nan")), torch.full_like(V, float("nan"))
        delta = torch.empty(Bn, nh, Lq, device=DEV)
        ops.attn_bwd(a, dO, dQ, dK, dV, delta)
        torch.cuda.synchronize()
        outs.append((O.clone(), dQ.clone(), dK.clone(), dV.clone()))
    for x, y in zip(*outs):
        assert torch.isfinite(x.float()).all() and torch.equal(x, y)
    # the bits against the mask probe: element (b, h, q, k) <-> word (k & 3) of tile (q >> 4, k >> 4), bit 16 * ((k & 15) >> 2) + (q & 15)
    Lkp = (Lk + 3) // 4 * 4
    mask = (ops.dropout_mask(Bn * nh * Lq * Lkp, p, 33, rng, DEV).view(Bn, nh, Lq, Lkp)[..., :Lk] != 0).cpu()
    w = bits.view(Bn, nh, (Lq + 15) // 16, (Lk + 15) // 16, 4).cpu()
    qi = torch.arange(Lq)[:, None].expand(Lq, Lk)
    ki = torch.arange(Lk)[None, :].expand(Lq, Lk)
    word = w[:, :, qi >> 4, ki >> 4, ki & 3]                                  # [Bn, nh, Lq, Lk]
    got = ((word >> (16 * ((ki & 15) >> 2) + (qi & 15))) & 1) != 0
    assert torch.equal(got, mask)
    assert 0.85 < mask.float().mean().item() < 0.95


# ---- ABI 6: top-p and unbounded top-k inside the sampling kernel ----------------------------------------
def _top_p_by_value(z, top_p):
    """The kernel's rule in torch (double): a token stays when the softmax mass of the STRICTLY larger logits is <= top_p.  Without
    ties this is utils/decoding_utils.py:22-34 (remove cumsum(softmax(sorted)) > top_p, shifted right by one)."""
    p = torch.softmax(z.double(), -1)
    sz, si = torch.sort(z.double(), descending=True, dim=-1)
    sp = p.gather(-1, si)
    front = torch.cumsum(sp, -1) - sp                                     # mass sorted in front of each position
    first = torch.ones_like(sz, dtype=torch.bool)
    first[:, 1:] = sz[:, 1:] != sz[:, :-1]                                # first position of every group of equal logits
    above = torch.cummax(torch.where(first, front, torch.full_like(front, -1.0)), -1).values
    keep = torch.zeros_like(first).scatter(-1, si, above <= top_p)
    return z.masked_fill(~keep, float("-inf")), above.gather(-1, si.argsort(-1))


def _assert_draws(got, z_filtered, u, what):
    from gst_visdial_amd import decoding
    prob = torch.softmax(z_filtered.double(), -1)
    want = decoding.draw_from_uniform(prob.float(), u).view(-1)
    assert (prob.gather(1, got[:, None]) > 0).all(), what                 # never a filtered token
    c = torch.cumsum(prob, -1)
    bad = 0
    for b in (got != want).nonzero().view(-1).tolist():
        lo, hi = sorted((int(got[b]), int(want[b])))
        on_step = (abs(float(c[b, hi - 1] - c[b, lo])) if hi - 1 >= lo else 0.0) < 1e-5 and \
            min(abs(float(c[b, lo]) - float(u[b])), abs(float(c[b, hi - 1]) - float(u[b]))) < 1e-5
        assert on_step, (what, b, int(got[b]), int(want[b]))
        bad += 1
    assert bad <= 1, what


@pytest.mark.parametrize("V", [97, 30522])
def test_sampling_kernel_top_p_and_wide_top_k_equal_the_torch_filters(V):
    """utils/decoding_utils.py:4-35 inside gstvd_sample_topk: top-p alone, behind a narrow and a wide top-k, k beyond the former
    limit of 64 and beyond the vocabulary; fp32 logits without ties: the kept set equals the reference filter's
    (decoding.batch_top_k_top_p_sampling = the reference's code path) wherever no token sits within 1e-6 of the top_p boundary,
    and the drawn ids equal the inverse-CDF draw over it."""
    from gst_visdial_amd import decoding, ops
    Bn = 12
    g = torch.Generator().manual_seed(1000 + V)
    for case, (top_k, top_p, temp) in enumerate([(0, 0.9, 1.0), (0, 0.5, 0.7), (0, 0.05, 1.0), (7, 0.9, 0.7), (50, 0.8, 1.3), (200, 0.0, 1.0),
                                                  (1000, 0.95, 0.8), (V + 5, 0.0, 1.0), (V + 5, 0.6, 1.0), (0, 1.0, 1.0), (65, 0.0, 1.0)]):
        logits = (torch.randn(Bn, V, generator=g) * 2.0).to(DEV)
        u = torch.rand(Bn, generator=g).clamp_min(1e-6).to(DEV)
        got = torch.full((Bn,), -1, dtype=torch.long, device=DEV)
        ops.sample_topk(logits, temp, top_k, u, got, None, top_p=top_p)
        z = logits / temp
        zk = decoding.batch_top_k_top_p_sampling(z, top_k=top_k, top_p=0.0)
        z_ref = decoding.batch_top_k_top_p_sampling(z, top_k=top_k, top_p=top_p)          # the reference's rule (sort + shifted cumsum)
        if 0.0 < top_p < 1.0:
            z_val, above = _top_p_by_value(zk, top_p)
            clear = ((above - top_p).abs() > 1e-6) | torch.isinf(zk)                        # rows whose boundary is not a rounding matter
            rows = clear.all(-1)
            assert rows.sum().item() >= Bn - 1, case
            assert torch.equal(torch.isinf(z_ref[rows]), torch.isinf(z_val[rows])), case    # by-value rule == reference rule (no ties)
        else:
            z_val = z_ref
        assert ((got >= 0) & (got < V)).all()
        _assert_draws(got, z_val, u, (case, top_k, top_p))
        if top_k > 0:
            assert (~torch.isinf(z_val)).sum(-1).max().item() <= min(top_k, V), case       # (no ties in fp32 noise: exactly <= k survive)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_sampling_kernel_top_p_keeps_equal_logits_together(dtype):
    """Logits with many exact ties (bf16 logits, quantised fp32): the kernel's kept set is the by-value rule's -- equal logits stay
    or go together (documented in include/gstvd_hip.h; the reference leaves the order inside a tie to torch.sort) -- and it always
    contains the reference rule's first kept token (the arg-max)."""
    from gst_visdial_amd import decoding, ops
    Bn, V = 10, 30522
    g = torch.Generator().manual_seed(7)
    for case, (top_k, top_p) in enumerate([(0, 0.9), (0, 0.3), (100, 0.7), (7, 0.5)]):
        logits = (torch.randn(Bn, V, generator=g) * 2.0)
        logits = ((logits * 8).round() / 8).to(DEV).to(dtype)                              # 1/8 steps: thousands of ties per row
        logits[3, :] = logits[3, 0]                                                        # a constant row
        u = torch.rand(Bn, generator=g).clamp_min(1e-6).to(DEV)
        got = torch.full((Bn,), -1, dtype=torch.long, device=DEV)
        ops.sample_topk(logits, 0.9, top_k, u, got, None, top_p=top_p)
        z = logits.float() / 0.9
        zk = decoding.batch_top_k_top_p_sampling(z, top_k=top_k, top_p=0.0)
        z_val, above = _top_p_by_value(zk, top_p)
        clear = ((above - top_p).abs() > 1e-6) | torch.isinf(zk)
        rows = clear.all(-1)
        assert rows.sum().item() >= Bn - 2, case
        sel = rows.nonzero().view(-1)
        _assert_draws(got[sel], z_val[sel], u[sel], (case, top_k, top_p))
        assert (~torch.isinf(z_val.gather(1, z.argmax(-1, keepdim=True)))).all()
