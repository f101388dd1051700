"""Per-kernel parity through the C ABI on a real MI355X: every HIP op against a plain PyTorch fp32
reference of the same op (fp32 mode: tight tolerance; bf16 mode: tolerance of one bf16 rounding of the
output scale).  Dropout is checked exactly by materialising the kernel's own counter-based mask."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def ops():
    from gst_visdial_amd import ops as o
    return o


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1.2e-2


def check(name, got, ref, dtype, mult=1.0):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), name + ": non-finite output"
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item() / scale
    assert err <= tol(dtype) * mult, "%s: rel-to-max error %.3e (scale %.3e, tol %.1e)" % (name, err, scale, tol(dtype) * mult)


def rnd(*shape, dtype=torch.float32, seed=0, s=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * s).to(DEV).to(dtype)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(200, 96, 64), (300, 256, 128), (111, 64, 320), (592, 1024, 256), (1024, 768, 192),
                                   (2816, 3072, 160), (3000, 2824, 104)])      # the last two take the 256x256x32 tile
def test_gemm_forward_bias(dtype, M, N, K):
    o = ops()
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, s=0.1)
    b = rnd(N, seed=3)
    y = torch.full((M, N), float("nan"), device=DEV, dtype=dtype)
    o.gemm(x, w, y, M, N, K, bias=b)
    check("gemm_nt", y, x.float() @ w.float().t() + b, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric_identity(dtype):
    """A = I with an asymmetric B catches transposed C writes (guide: always A=I-check with asymmetric B)."""
    o = ops()
    M = N = K = 64
    a = torch.eye(M, device=DEV, dtype=dtype)
    w = (torch.arange(N * K, device=DEV).float().reshape(N, K) % 13 - 6).to(dtype)   # exact small ints
    y = torch.empty(M, N, device=DEV, dtype=dtype)
    o.gemm(a, w, y, M, N, K)
    assert torch.equal(y.float(), w.float().t())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(200, 64, 96), (300, 128, 256), (592, 256, 1024), (3000, 2824, 200)])
def test_gemm_dgrad_kmajor_b(dtype, M, N, K):
    """dx[M,N] = dy[M,K] @ W[K,N]  with W stored [K,N] (k-major B)."""
    o = ops()
    dy, w = rnd(M, K, dtype=dtype, seed=4), rnd(K, N, dtype=dtype, seed=5, s=0.1)
    r = rnd(M, N, dtype=dtype, seed=6)
    dx = torch.empty(M, N, device=DEV, dtype=dtype)
    o.gemm(dy, w, dx, M, N, K, b_km=True, addend=r)
    check("gemm_nn_add", dx, dy.float() @ w.float() + r.float(), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,N,K", [(111, 64, 96), (300, 128, 256), (4100, 256, 128), (333, 3072, 2568)])
def test_gemm_wgrad_accumulate(dtype, rows, N, K):
    """dW[N,K] (+)= dy[rows,N]^T @ x[rows,K]; fp32 output, both operands k-major, contraction not a tile multiple."""
    o = ops()
    dy, x = rnd(rows, N, dtype=dtype, seed=7), rnd(rows, K, dtype=dtype, seed=8)
    dw = rnd(N, K, seed=9)
    ref = dw + dy.float().t() @ x.float()
    o.gemm(dy, x, dw, N, K, rows, a_km=True, b_km=True, addend=dw)
    check("gemm_tn_acc", dw, ref, dtype, mult=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_gelu_epilogues(dtype):
    o = ops()
    M, N, K = 150, 128, 64
    x, w, b = rnd(M, K, dtype=dtype, seed=10), rnd(N, K, dtype=dtype, seed=11, s=0.2), rnd(N, seed=12)
    u = torch.empty(M, N, device=DEV, dtype=dtype)
    a = torch.empty(M, N, device=DEV, dtype=dtype)
    o.gemm(x, w, a, M, N, K, bias=b, aux=u, epi=o.EPI_GELU)
    uref = (x.float() @ w.float().t() + b).requires_grad_(True)
    aref = torch.nn.functional.gelu(uref)
    aref.backward(torch.ones_like(aref))
    check("gelu_out", a, aref.detach(), dtype)
    check("gelu_deriv", u, uref.grad, dtype)          # aux keeps gelu'(pre-activation) for the backward epilogue
    # dgelu: d_u = (d_a @ W2) * gelu'(u)
    da, w2 = rnd(M, K, dtype=dtype, seed=13), rnd(K, N, dtype=dtype, seed=14, s=0.2)
    du = torch.empty(M, N, device=DEV, dtype=dtype)
    o.gemm(da, w2, du, M, N, K, b_km=True, aux=u, epi=o.EPI_DGELU)
    check("dgelu", du, (da.float() @ w2.float()) * uref.grad, dtype)


@pytest.mark.parametrize("case", [
    # M, N, K, padded leading dimensions, epilogue, out dtype, batch -- all take the full-line (64-deep fill) NT 256-tile kernel
    (4096, 3072, 768, 0, "gelu", torch.bfloat16, 1),      # FFN-up: 192-wide tiles, one per CU
    (4096, 2304, 768, 0, "bias", torch.bfloat16, 1),      # QKV
    (3000, 2824, 1024, 24, "add", torch.bfloat16, 1),     # ragged rows / columns, padded leading dimensions, K = 1024
    (2048, 4096, 128, 0, "bias", torch.bfloat16, 1),      # two fills only (the prologue's), 256-wide tiles
    (3000, 2824, 3072, 8, "none", torch.float32, 1),      # fp32 output: tile-wise epilogue; 48 fills
    (1024, 1280, 192, 0, "bias", torch.bfloat16, 6),      # batched (120 tiles), K = 3 fills
])
def test_gemm_nt_full_line_tile(case):
    """Round 4's pc_tile256_nt64 (row-major A and B, K % 64 == 0, 120..512 tiles): results against torch fp32 on every epilogue
    / tail / stride combination the dispatcher can send it, and the library really dispatches these shapes to it."""
    o = ops()
    M, N, K, pad, epi, odt, Bn = case
    bf = torch.bfloat16
    x = rnd(Bn * M, K + pad, dtype=bf, seed=70)[:, :K]
    w = rnd(Bn * N, K + pad, dtype=bf, seed=71, s=0.1)[:, :K]
    b = rnd(N, seed=72)
    y = torch.full((Bn * M, N + pad), float("nan"), device=DEV, dtype=odt)[:, :N]
    kw = dict(lda=K + pad, ldb=K + pad, ldc=N + pad)
    if Bn > 1:
        kw.update(batch=Bn, sA=M * (K + pad), sB=N * (K + pad), sC=M * (N + pad))
    xf, wf = x.float().view(Bn, M, K), w.float().view(Bn, N, K)
    ref = torch.einsum("bmk,bnk->bmn", xf, wf)
    aux = None
    if epi == "bias":
        o.gemm(x, w, y, M, N, K, bias=b, **kw); ref = ref + b
    elif epi == "gelu":
        aux = torch.empty(M, N, device=DEV, dtype=bf)
        o.gemm(x, w, y, M, N, K, bias=b, aux=aux, epi=o.EPI_GELU, **kw)
        u = (ref + b).requires_grad_(True)
        g = torch.nn.functional.gelu(u); g.backward(torch.ones_like(g))
        ref = g.detach()
        check("nt64_gelu_deriv", aux, u.grad.view(M, N), bf)
    elif epi == "add":
        r = rnd(M, N + pad, dtype=bf, seed=73)[:, :N]
        o.gemm(x, w, y, M, N, K, addend=r, ldadd=N + pad, **kw); ref = ref + r.float()
    else:
        o.gemm(x, w, y, M, N, K, **kw)
    check("nt64", y.reshape(Bn, M, N) if pad == 0 else torch.stack([y[i * M:(i + 1) * M] for i in range(Bn)]), ref, bf)
    with o.Profiler() as prof:                 # which device symbol did the library launch for this descriptor?
        o.gemm(x, w, y, M, N, K, **kw)
    launched = [k for k in prof.summary() if k.startswith("gemm:")]
    assert len(launched) == 1 and "nt64" in launched[0], launched


@pytest.mark.parametrize("lay", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(400, 768, 3072), (100, 200, 1992), (64, 64, 30528), (37, 132, 2304)])
def test_gemm_splitk_matches_single_pass(lay, shape):
    """Skinny, deep bf16 problems take the split-K kernel (gstvd_gemm_splitk); same result as one pass, epilogue included,
    and bit-identical from run to run (partials are added in split order, not arrival order)."""
    o = ops()
    M, N, K = shape
    a_km, b_km = (lay == "tn"), (lay in ("nn", "tn"))
    if a_km and M % 8:
        M = (M + 7) // 8 * 8
    if b_km and N % 8:
        N = (N + 7) // 8 * 8
    N = (N + 3) // 4 * 4
    A = rnd(*((K, M) if a_km else (M, K)), dtype=torch.bfloat16, seed=31, s=0.5)
    B = rnd(*((K, N) if b_km else (N, K)), dtype=torch.bfloat16, seed=32, s=0.5)
    bias, add = rnd(N, seed=33), rnd(M, N, dtype=torch.bfloat16, seed=34)
    assert o.splitk_plan(o.BF16, M, N, K, 1, a_km, b_km) > 1
    outs = []
    for rep in range(3):
        Cs = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        o.gemm(A, B, Cs, M, N, K, a_km=a_km, b_km=b_km, bias=bias, addend=add)
        outs.append(Cs)
    ref = (A.float().t() if a_km else A.float()) @ (B.float() if b_km else B.float().t()) + bias + add.float()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    check("splitk", outs[0], ref, torch.bfloat16)
    old, o.SPLITK = o.SPLITK, 0
    try:
        C1 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        o.gemm(A, B, C1, M, N, K, a_km=a_km, b_km=b_km, bias=bias, addend=add)
    finally:
        o.SPLITK = old
    # fp32 accumulation in a different order: identical up to the final bf16 rounding
    assert (outs[0].float() - C1.float()).abs().max() <= 2.0 ** -7 * ref.abs().max()


def test_dropout_masks_of_a_site_larger_than_2p24_draws_do_not_repeat():
    """a site with more than 2^24 draws (2^25 elements) got exactly duplicated mask pairs at XOR distance 0x01000100
    of the pair counter.  Device masks of a 2^26-element site: the keep bits at that distance must be independent
    (agreement p^2 + (1-p)^2), and the first 2^25 elements are the stream the smaller sites have always had."""
    o = ops()
    rng = o.Rng(DEV, seed=77)
    n, p = 1 << 26, 0.25
    keep = o.dropout_mask(n, p, 5, rng, DEV) != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 1e-3
    e = torch.arange(1 << 22, device=DEV) + (1 << 23)                   # some elements of the first 2^24-draw block
    partner = (((e >> 1) ^ 0x01000100) << 1) | (e & 1)                  # the element whose pair counter differs in bits 8 and 24
    agree = (keep[e] == keep[partner]).float().mean().item()
    assert abs(agree - (p * p + (1 - p) * (1 - p))) < 5e-3, agree      # 0.625; the defect gave 1.0
    small = o.dropout_mask(1 << 20, p, 5, rng, DEV) != 0
    assert torch.equal(small, keep[: 1 << 20])


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_and_dropout(dtype):
    o = ops()
    Bn, M, N, K = 3, 37, 64, 96
    x, w = rnd(Bn, M, K, dtype=dtype, seed=15), rnd(N, K, dtype=dtype, seed=16, s=0.1)
    b = rnd(N, seed=17)
    S = M + 11
    y = torch.zeros(Bn, S, N, device=DEV, dtype=dtype)
    rng = o.Rng(DEV, seed=1234)
    o.gemm(x, w, y[:, 5:], M, N, K, bias=b, batch=Bn, sA=M * K, sC=S * N, lda=K, ldc=N, drop_p=0.1, site=7, rng=rng)
    mask = o.dropout_mask(Bn * M * N, 0.1, 7, rng, DEV).view(Bn, M, N)
    keep = (mask != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.02, keep
    assert torch.all((mask == 0) | ((mask - 1 / 0.9).abs() < 1e-6))
    ref = (x.float() @ w.float().t() + b) * mask
    check("gemm_batched_dropout", y[:, 5:5 + M], ref, dtype)
    assert torch.all(y[:, :5] == 0) and torch.all(y[:, 5 + M:] == 0)


# ------------------------------------------------------------------------------------------ LayerNorm
def ln_ref(h, gamma, beta, eps=1e-12):
    u = h.mean(-1, keepdim=True)
    s = (h - u).pow(2).mean(-1, keepdim=True)
    return gamma * ((h - u) / torch.sqrt(s + eps)) + beta


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,H", [(70, 64), (45, 96), (130, 768), (50, 1024)])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_ln_resid_fwd_bwd(dtype, M, H, p):
    o = ops()
    x, res = rnd(M, H, dtype=dtype, seed=20), rnd(M, H, dtype=dtype, seed=21)
    gamma, beta = 1 + 0.1 * rnd(H, seed=22), 0.1 * rnd(H, seed=23)
    dy = rnd(M, H, dtype=dtype, seed=24)
    rng = o.Rng(DEV, seed=99)
    y = torch.empty(M, H, device=DEV, dtype=dtype)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    kw = dict(mode=o.LN_RESID, dtype=o.dt(x), M=M, H=H, gamma=gamma, beta=beta, mean=mean, rstd=rstd, eps=1e-12,
              x=x, res=res, y=y, p_pre=p, site_pre=3, rng=rng)
    o.ln_fwd(**kw)
    mask = o.dropout_mask(M * H, p, 3, rng, DEV).view(M, H) if p > 0 else torch.ones(M, H, device=DEV)
    xr, rr = x.float().requires_grad_(True), res.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yref = ln_ref(xr * mask + rr, gr, br)
    check("ln_fwd", y, yref, dtype)
    yref.backward(dy.float())
    nblk = o.ln_bwd_blocks(M)
    partial = torch.empty(nblk, 3, H, device=DEV)
    dres = torch.empty(M, H, device=DEV, dtype=dtype)
    dx = torch.empty(M, H, device=DEV, dtype=dtype) if p > 0 else dres
    o.ln_bwd(kw, dy, partial, dres=dres, dx=dx)
    dg, db, dbias = torch.zeros(H, device=DEV), torch.ones(H, device=DEV), torch.empty(H, device=DEV)
    o.colsum_partials(partial, nblk, 3, H, dg, db, dbias, accumulate=False)
    check("ln_dres", dres, rr.grad, dtype, 2.0)
    check("ln_dx", dx, xr.grad, dtype, 2.0)
    check("ln_dgamma", dg, gr.grad, dtype, 3.0)
    check("ln_dbeta", db, br.grad, dtype, 3.0)
    check("ln_dbias", dbias, xr.grad.sum(0), dtype, 3.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_ln_embed_fwd_bwd(dtype):
    o = ops()
    Bn, T, H, V = 3, 24, 64, 50
    M = Bn * T
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (Bn, T), generator=g).to(DEV)
    segs = torch.randint(0, 4, (Bn, T), generator=g).to(DEV)      # includes extension ids 2,3
    word, pos, tt, tte = rnd(V, H, seed=30), rnd(40, H, seed=31), rnd(2, H, seed=32), rnd(10, H, seed=33)
    gamma, beta = 1 + 0.1 * rnd(H, seed=34), 0.1 * rnd(H, seed=35)
    y = torch.empty(M, H, device=DEV, dtype=dtype)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    rng = o.Rng(DEV, seed=5)
    kw = dict(mode=o.LN_EMBED, dtype=o.dt(y), M=M, H=H, gamma=gamma, beta=beta, mean=mean, rstd=rstd, eps=1e-12, y=y,
              ids=ids.view(-1), segs=segs.view(-1), T=T, type_vocab=2, word=word, pos=pos, tt=tt, tt_ext=tte,
              p_post=0.3, site_post=11, rng=rng)
    o.ln_fwd(**kw)
    mask = o.dropout_mask(M * H, 0.3, 11, rng, DEV).view(M, H)
    ws = [t.clone().requires_grad_(True) for t in (word, pos, tt, tte, gamma, beta)]
    typ = torch.where((segs < 2)[..., None], ws[2][segs.clamp(max=1)], ws[3][(segs - 2).clamp(min=0)])
    h = ws[0][ids] + ws[1][:T][None] + typ
    yref = ln_ref(h, ws[4], ws[5]).view(M, H) * mask
    check("embed_fwd", y, yref, dtype)
    dy = rnd(M, H, dtype=dtype, seed=36)
    yref.backward(dy.float())
    nblk = o.ln_bwd_blocks(M)
    partial = torch.empty(nblk, 4, H, device=DEV)
    dws = [torch.zeros_like(t) for t in (word, pos, tt, tte)]
    o.ln_bwd(kw, dy, partial, dword=dws[0], dpos=dws[1], dtt=dws[2], dtt_ext=dws[3])
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    batch = o.ColsumBatch(DEV)
    batch.add(partial.view(-1), (dg, db, dws[2][0]), nblk, 4 * H, H, 3, (False, False, True))
    batch.add(partial.view(-1)[3 * H:], (dws[2][1], None, None), nblk, 4 * H, H, 1, (True, False, False))
    batch.flush()
    for name, got, ref in zip(("dword", "dpos", "dtt", "dtt_ext"), dws, ws[:4]):
        check("embed_" + name, got, ref.grad, dtype, 3.0)
    check("embed_dgamma", dg, ws[4].grad, dtype, 3.0)
    check("embed_dbeta", db, ws[5].grad, dtype, 3.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_ln_image_fwd_bwd(dtype):
    o = ops()
    M, H = 21, 96
    x = rnd(M, H, dtype=dtype, seed=40)
    loc = torch.rand(M, 5, device=DEV)
    wl, bl = rnd(H, 5, seed=41, s=0.3), rnd(H, seed=42, s=0.1)
    gamma, beta = 1 + 0.1 * rnd(H, seed=43), 0.1 * rnd(H, seed=44)
    y = torch.empty(M, H, device=DEV, dtype=dtype)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    kw = dict(mode=o.LN_IMAGE, dtype=o.dt(x), M=M, H=H, gamma=gamma, beta=beta, mean=mean, rstd=rstd, eps=1e-12,
              x=x, y=y, loc=loc, w_loc=wl, b_loc=bl)
    o.ln_fwd(**kw)
    xr, wr, br = x.float().requires_grad_(True), wl.clone().requires_grad_(True), bl.clone().requires_grad_(True)
    yref = ln_ref(xr + loc @ wr.t() + br, gamma, beta)
    check("image_fwd", y, yref, dtype)
    dy = rnd(M, H, dtype=dtype, seed=45)
    yref.backward(dy.float())
    nblk = o.ln_bwd_blocks(M)
    partial = torch.empty(nblk, 3, H, device=DEV)
    dh = torch.empty(M, H, device=DEV, dtype=dtype)
    o.ln_bwd(kw, dy, partial, dres=dh)
    dbl = torch.empty(H, device=DEV)
    o.colsum_partials(partial, nblk, 3, H, None, None, dbl, accumulate=False)
    check("image_dh", dh, xr.grad, dtype, 2.0)
    check("image_dbloc", dbl, br.grad, dtype, 3.0)
    dwl = torch.empty(H, 5, device=DEV)
    o.locgrad(dh, loc, M, H, dwl, accumulate=False)
    check("image_dwloc", dwl, wr.grad, dtype, 3.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum(dtype):
    o = ops()
    M, N = 333, 2304
    x = rnd(M, N, dtype=dtype, seed=50)
    out = torch.ones(N, device=DEV)
    scratch = torch.empty(((M + 63) // 64) * N, device=DEV)
    o.colsum(x, M, N, out, scratch, accumulate=True)
    check("colsum", out, 1 + x.float().sum(0), dtype)


# ------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, key_mask, causal, neg, scale, dmask):
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    Lq, Lk = q.shape[1], k.shape[1]
    allowed = (key_mask != 0)[:, None, None, :].expand(-1, 1, Lq, -1)
    if causal:
        i = torch.arange(Lq, device=q.device)[:, None]
        j = torch.arange(Lk, device=q.device)[None, :]
        allowed = allowed & (j <= i)[None, None]
    s = s + torch.where(allowed, torch.zeros((), device=q.device), torch.full((), neg, device=q.device))
    p = torch.softmax(s, -1)
    lse = torch.logsumexp(s, -1)
    if dmask is not None:
        p = p * dmask
    return torch.einsum("bhqk,bkhd->bqhd", p, v), lse


ATTN_CASES = [
    # B, nh, Lq, Lk, d, causal, neg, p, fused
    (2, 3, 40, 70, 32, False, -10000.0, 0.0, False),
    (2, 2, 24, 24, 32, False, -10000.0, 0.1, True),     # tiny text self-attention
    (1, 12, 256, 256, 64, False, -10000.0, 0.1, True),  # text self-attention, full size
    (2, 8, 37, 37, 128, False, -10000.0, 0.1, True),    # vision self-attention
    (2, 8, 256, 37, 128, False, -10000.0, 0.1, False),  # co-attention: text queries, vision keys
    (2, 8, 37, 256, 128, False, -10000.0, 0.1, False),  # co-attention: vision queries, text keys
    (3, 12, 25, 25, 64, True, -10000.0, 0.1, True),     # decoder causal self-attention
    (2, 12, 25, 293, 64, False, -1e9, 0.1, False),      # decoder cross-attention
    (2, 2, 9, 31, 32, True, -10000.0, 0.0, False),      # ragged tiny
    (2, 4, 65, 65, 64, False, -10000.0, 0.1, True),     # one row / one key spills into a second 64-chunk
    (1, 2, 130, 128, 64, False, -10000.0, 0.0, False),  # key count an exact multiple of the chunk
    (2, 2, 100, 100, 64, True, -10000.0, 0.1, True),    # causal across chunk boundaries
    (1, 4, 1, 200, 128, False, -1e9, 0.0, False),       # a single query (decode step shape)
    (44, 12, 256, 256, 64, False, -10000.0, 0.1, True), # 34.6 M score elements = 17.3 M dropout draws: past 2^24, where the counter's
                                                        # top byte used to drop out of the hash -- kernel masks == probe masks
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_fwd_bwd(dtype, case):
    o = ops()
    Bn, nh, Lq, Lk, d, causal, neg, p, fused = case
    H = nh * d
    if fused:
        qkv = rnd(Bn * Lq, 3 * H, dtype=dtype, seed=60)
        Q, K, V = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    else:
        Q, K, V = rnd(Bn * Lq, H, dtype=dtype, seed=61), rnd(Bn * Lk, H, dtype=dtype, seed=62), rnd(Bn * Lk, H, dtype=dtype, seed=63)
    km = torch.ones(Bn, Lk, device=DEV)
    km[0, Lk - Lk // 3:] = 0
    if Bn > 1:
        km[1, Lk // 2] = 0
    O = torch.full((Bn * Lq, H), float("nan"), device=DEV, dtype=dtype)
    LSE = torch.empty(Bn, nh, Lq, device=DEV)
    rng = o.Rng(DEV, seed=77)
    a = o.attn_desc(Q, K, V, O, LSE, km, Bn, nh, Lq, Lk, d, causal=causal, mask_neg=neg, drop_p=p, site=21, rng=rng)
    o.attn_fwd(a)
    Lkp = (Lk + 3) // 4 * 4
    dmask = None
    if p > 0:
        dmask = o.dropout_mask(Bn * nh * Lq * Lkp, p, 21, rng, DEV).view(Bn, nh, Lq, Lkp)[..., :Lk]
    qr = Q.float().reshape(Bn, Lq, nh, d).clone().requires_grad_(True)
    kr = K.float().reshape(Bn, Lk, nh, d).clone().requires_grad_(True)
    vr = V.float().reshape(Bn, Lk, nh, d).clone().requires_grad_(True)
    oref, lse_ref = attn_ref(qr, kr, vr, km, causal, neg, 1.0 / math.sqrt(d), dmask)
    check("attn_out", O.view(Bn, Lq, nh, d), oref, dtype, 2.0)
    check("attn_lse", LSE, lse_ref, dtype, 2.0)
    dO = rnd(Bn * Lq, H, dtype=dtype, seed=64)
    oref.backward(dO.float().view(Bn, Lq, nh, d))
    if fused:
        dqkv = torch.full_like(qkv, float("nan"))
        dQ, dK, dV = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]
    else:
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    delta = torch.empty(Bn, nh, Lq, device=DEV)
    o.attn_bwd(a, dO, dQ, dK, dV, delta)
    check("attn_dq", dQ.reshape(Bn, Lq, nh, d), qr.grad, dtype, 4.0)
    check("attn_dk", dK.reshape(Bn, Lk, nh, d), kr.grad, dtype, 4.0)
    check("attn_dv", dV.reshape(Bn, Lk, nh, d), vr.grad, dtype, 4.0)


# ------------------------------------------------------------------------------------------ loss & misc
@pytest.mark.parametrize("dtype", DTYPES)
def test_cross_entropy_fwd_bwd_and_scores(dtype):
    o = ops()
    rows, U, V, ld = 6, 9, 322, 384
    M = rows * U
    logits = torch.zeros(M, ld, device=DEV, dtype=dtype)
    logits[:, :V] = rnd(M, V, dtype=dtype, seed=70, s=2.0)
    g = torch.Generator().manual_seed(9)
    labels = torch.randint(1, V, (M,), generator=g).to(DEV)
    labels[::3] = 0
    row_loss, lse, stats = torch.empty(M, device=DEV), torch.empty(M, device=DEV), torch.empty(2, device=DEV)
    o.ce_fwd(logits, labels, M, V, row_loss, lse, stats)
    lr = logits[:, :V].float().requires_grad_(True)
    ref_none = torch.nn.functional.cross_entropy(lr, labels, ignore_index=0, reduction="none")
    ref_mean = torch.nn.functional.cross_entropy(lr, labels, ignore_index=0)
    check("ce_rows", row_loss, ref_none, torch.float32, 5.0)
    assert abs((stats[0] / stats[1]).item() - ref_mean.item()) < 1e-4 * max(1.0, abs(ref_mean.item()))
    assert stats[1].item() == (labels != 0).sum().item()
    ref_mean.backward()
    dl = torch.full((M, ld), float("nan"), device=DEV, dtype=dtype)
    gs = torch.ones(1, device=DEV)
    o.ce_bwd(logits, labels, lse, stats, gs, True, M, V, dl)
    check("ce_dlogits", dl[:, :V], lr.grad, dtype, 2.0)
    assert torch.all(dl[:, V:] == 0)
    ids = torch.randint(1, V, (rows, U), generator=g).to(DEV)
    ids[:, 6:] = 0
    scores = torch.empty(rows, device=DEV)
    o.answer_scores(logits, lse, ids, rows, U, scores)
    lp = torch.log_softmax(logits[:, :V].float(), -1).view(rows, U, V)
    tgt = torch.zeros_like(ids)
    tgt[:, :-1] = ids[:, 1:]
    ref = (torch.gather(lp, -1, tgt[..., None]).squeeze(-1) * (tgt != 0).float()).sum(-1)
    check("answer_scores", scores, ref, torch.float32, 10.0)


def test_cast_roundtrip_and_adamw():
    o = ops()
    n = 5003
    x = rnd(n + 5, seed=80)[:n + 5]
    xb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    o.cast(x[:n], xb)
    assert torch.equal(xb, x[:n].to(torch.bfloat16))
    # AdamW, pytorch_transformers 1.2.0 semantics
    p0, g0 = rnd(n, seed=81), rnd(n, seed=82, s=0.1)
    m0, v0 = rnd(n, seed=83, s=0.01), rnd(n, seed=84, s=0.01).abs()
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    seg_end = torch.tensor([1000, 3001, n], device=DEV)
    hp = torch.tensor([2e-5, 0.01, 1e-3, 0.0, 5e-4, 0.1], device=DEV)
    step = torch.tensor([3.0], device=DEV)
    o.adamw(p, g0, m, v, shadow, seg_end, hp, step)
    lr = torch.empty(n, device=DEV)
    wd = torch.empty(n, device=DEV)
    for (a, b), l_, w_ in zip(((0, 1000), (1000, 3001), (3001, n)), (2e-5, 1e-3, 5e-4), (0.01, 0.0, 0.1)):
        lr[a:b], wd[a:b] = l_, w_
    mr = m0 * 0.9 + 0.1 * g0
    vr = v0 * 0.999 + 0.001 * g0 * g0
    ss = lr * math.sqrt(1 - 0.999 ** 3) / (1 - 0.9 ** 3)
    pr = p0 - ss * mr / (vr.sqrt() + 1e-6)
    pr = pr - lr * wd * pr
    check("adamw_p", p, pr, torch.float32, 5.0)
    check("adamw_m", m, mr, torch.float32)
    check("adamw_v", v, vr, torch.float32)
    assert torch.equal(shadow, p.to(torch.bfloat16))
    # slice [begin, end) fed from a bf16 copy of the slice's gradients (the compressed all-reduce path, N>1):
    # identical to the fp32-gradient kernel run on the bf16-rounded gradients; elements outside the slice untouched
    lo, hi = 1000, 4004
    gb = g0[lo:hi].to(torch.bfloat16)
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    sa, sb = torch.zeros_like(shadow), torch.zeros_like(shadow)
    gq = g0.clone(); gq[lo:hi] = gb.float()
    o.adamw(pa, gq, ma, va, sa, seg_end, hp, step, begin=lo, end=hi)
    o.adamw(pb, gb, mb, vb, sb, seg_end, hp, step, begin=lo, end=hi, grad_origin=lo)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(sa, sb)
    assert torch.equal(pb[:lo], p0[:lo]) and torch.equal(pb[hi:], p0[hi:])


def test_gemm_grouped_matches_individual_launches():
    """The deferred weight-gradient path: several dW = dy^T x problems (different shapes, one accumulating) in one launch."""
    o = ops()
    grp = o.GemmGroup(DEV, a_km=True, b_km=True)
    refs = []
    for i, (rows, N, K, acc) in enumerate([(333, 128, 256, False), (592, 1024, 256, True), (100, 64, 96, False), (4096, 768, 768, False)]):
        dy, x = rnd(rows, N, dtype=torch.bfloat16, seed=90 + i), rnd(rows, K, dtype=torch.bfloat16, seed=95 + i)
        dw = rnd(N, K, seed=99 + i)
        ref = dy.float().t() @ x.float() + (dw if acc else 0)
        grp.add(dy, x, dw, N, K, rows, acc)
        refs.append((dw, ref))
    grp.flush()
    for i, (dw, ref) in enumerate(refs):
        check("gemm_grouped_%d" % i, dw, ref, torch.bfloat16, 2.0)


def test_gemm_grouped_column_sums():
    """GSTVD_EPI_COLSUM: the grouped weight-gradient launch also returns the bias gradient sum_rows dy[row][:] (overwrite and
    accumulate forms; ragged row counts and widths below / across the 256 tile; a problem without the flag in the same launch)."""
    o = ops()
    grp = o.GemmGroup(DEV, a_km=True, b_km=True)
    refs = []
    cases = [(333, 128, 256, False, "set"), (592, 1024, 256, True, "acc"), (100, 64, 96, False, "acc"), (4096, 768, 768, False, "set"),
             (77, 300, 520, False, None), (1, 264, 32, False, "set")]
    for i, (rows, N, K, acc, cs) in enumerate(cases):
        dy, x = rnd(rows, N, dtype=torch.bfloat16, seed=190 + i), rnd(rows, K, dtype=torch.bfloat16, seed=195 + i)
        assert grp.colsum_capable(dy)
        dw, gb = rnd(N, K, seed=199 + i), rnd(N, seed=205 + i)
        ref = dy.float().t() @ x.float() + (dw if acc else 0)
        gb_ref = gb.clone() if cs is None else dy.double().sum(0).float() + (gb if cs == "acc" else 0)
        if cs is None:
            grp.add(dy, x, dw, N, K, rows, acc)
        else:
            grp.add(dy, x, dw, N, K, rows, acc, colsum_out=gb, colsum_acc=(cs == "acc"))
        refs.append((dw, ref, gb, gb_ref, rows))
    grp.flush()
    for i, (dw, ref, gb, gb_ref, rows) in enumerate(refs):
        check("gemm_grouped_cs_%d" % i, dw, ref, torch.bfloat16, 2.0)
        # fp32 sums of bf16 values: only the summation order differs from the float64 reference
        assert (gb - gb_ref).abs().max().item() <= 1e-5 * max(1.0, rows ** 0.5) * max(1.0, gb_ref.abs().max().item()), i
    # replays of the same group (the captured train step) keep working: second launch, accumulate form adds again
    dw, ref, gb, gb_ref, rows = refs[3]
    before = gb.clone()
    grp2 = o.GemmGroup(DEV, a_km=True, b_km=True)
    dy = rnd(4096, 768, dtype=torch.bfloat16, seed=193)
    grp2.add(dy, rnd(4096, 768, dtype=torch.bfloat16, seed=198), dw, 768, 768, 4096, False, colsum_out=gb, colsum_acc=True)
    grp2.flush()
    assert torch.allclose(gb, before + dy.float().sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nh,d,Lk", [(12, 64, 1), (12, 64, 19), (12, 64, 293), (3, 64, 1100), (8, 128, 37), (8, 128, 300), (4, 32, 77)])
def test_single_query_attention_kernel(dtype, nh, d, Lk):
    """The decode step's attention (Lq = 1: csrc/attention.hip attn_decode_kernel): one key up to more keys than a workgroup
    scores in one round, every head width, masked keys, a strided KV cache and keys / values shared by groups of rows; LSE."""
    o = ops()
    Bn, G, Umax = 6, 3, Lk + 5
    H = nh * d
    q = rnd(Bn, H, dtype=dtype, seed=210)
    kc, vc = rnd(2 * Umax, H, dtype=dtype, seed=211), rnd(2 * Umax, H, dtype=dtype, seed=212)     # two K/V rows of Umax slots each
    km = torch.ones(2, Lk, device=DEV)
    if Lk > 4:
        km[1, Lk // 2:] = 0
        km[0, 1] = 0
    out = torch.full((Bn, H), float("nan"), device=DEV, dtype=dtype)
    lse = torch.empty(Bn, nh, 1, device=DEV)
    a = o.attn_desc(q, kc, vc, out, lse, km, Bn, nh, 1, Lk, d, mask_neg=-10000.0, kv_group=G, kv_bstride=Umax)
    o.attn_fwd(a)
    rep = lambda t: t.repeat_interleave(G, dim=0)
    k4 = kc.view(2, Umax, nh, d)[:, :Lk].float()
    v4 = vc.view(2, Umax, nh, d)[:, :Lk].float()
    ref, lref = attn_ref(q.float().view(Bn, 1, nh, d), rep(k4), rep(v4), rep(km), False, -10000.0, 1 / math.sqrt(d), None)
    check("attn_decode", out.view(Bn, 1, nh, d), ref, dtype, 2.0)
    assert (lse.view(Bn, nh, 1) - lref).abs().max().item() <= (1e-4 if dtype == torch.float32 else 5e-2)


@pytest.mark.parametrize("M,N,K", [(16, 768, 768), (16, 2304, 768), (5, 136, 200), (16, 3072, 1024), (3, 30528, 768)])
def test_decode_layernorm_folded_into_the_linear(M, N, K):
    """gstvd_gemv_ln: C = epi(LN(A) . B^T) for the decode step's rows -- against LayerNorm (fp32 arithmetic on the bf16 rows,
    bf16 result) followed by the skinny GEMM; the normalised rows come back through y_out; bias / residual add / GELU."""
    o = ops()
    x = rnd(M, K, dtype=torch.bfloat16, seed=60, s=1.7) + 0.3
    w, b = rnd(N, K, dtype=torch.bfloat16, seed=61, s=0.1), rnd(N, seed=62)
    gam, bet = rnd(K, seed=63) * 0.2 + 1.0, rnd(K, seed=64) * 0.1
    yn = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet, 1e-12).to(torch.bfloat16)
    ref = yn.float() @ w.float().t() + b
    big = torch.zeros(M, 3, N, device=DEV, dtype=torch.bfloat16)
    c, yo = big[:, 1], torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
    o.gemv_ln(x, w, c, M, N, K, gam, bet, 1e-12, y_out=yo, bias=b)
    # the normalised rows: bf16 roundings of the same fp32 values (an ulp where the two roundings straddle a tie)
    assert (yo.float() - yn.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, yn.float().abs().max().item())
    check("gemv_ln", c.contiguous(), ref, torch.bfloat16, 2.0)
    assert float(big[:, 0].abs().max()) == 0.0 and float(big[:, 2].abs().max()) == 0.0
    r = rnd(M, N, seed=65)                                           # the addend has the OUTPUT's type (fp32 here), as in gstvd_gemm
    c2 = torch.empty(M, N, device=DEV, dtype=torch.float32)
    o.gemv_ln(x, w, c2, M, N, K, gam, bet, 1e-12, bias=b, addend=r)
    check("gemv_ln_add_f32out", c2, ref + r, torch.bfloat16, 2.0)
    rb = r.to(torch.bfloat16)
    c3 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o.gemv_ln(x, w, c3, M, N, K, gam, bet, 1e-12, bias=b, addend=rb)
    check("gemv_ln_add_bf16out", c3, ref + rb.float(), torch.bfloat16, 2.0)
    a, u = torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o.gemv_ln(x, w, a, M, N, K, gam, bet, 1e-12, bias=b, aux=u, epi=o.EPI_GELU)
    check("gemv_ln_gelu", a, torch.nn.functional.gelu(ref), torch.bfloat16, 2.0)
    with pytest.raises(Exception):                                   # rows that do not fit the kernel are refused, not mangled
        o.gemv_ln(rnd(17, K, dtype=torch.bfloat16, seed=1), w, torch.empty(17, N, device=DEV, dtype=torch.bfloat16), 17, N, K, gam, bet, 1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("V", [97, 30522])
def test_sample_topk_matches_the_torch_filters_and_inverse_cdf(dtype, V):
    """gstvd_sample_topk vs the torch-op form of the same step (decoding.py: temperature, ban, top-k with ties, softmax,
    inverse-CDF draw): same ids, except where a uniform falls within rounding distance of a CDF step."""
    from gst_visdial_amd import decoding
    o = ops()
    Bn = 16
    g = torch.Generator().manual_seed(V)
    for case, (top_k, temp, with_ban) in enumerate([(0, 1.0, False), (1, 0.7, True), (7, 0.7, False), (7, 0.7, True), (40, 1.3, True), (64, 2.0, False)]):
        logits = (torch.randn(Bn, V, generator=g) * 2.5).to(DEV).to(dtype)
        logits[:, ::3] = (logits[:, ::3].float() * 4).round().div(4).to(dtype)      # plenty of exact ties
        logits[1, :] = logits[1, 0]                                                  # a constant row: every token ties
        banned = None
        if with_ban:
            banned = torch.zeros(Bn, V + 1, dtype=torch.bool, device=DEV)
            banned[:, torch.randint(0, V, (V // 5,), generator=g).to(DEV)] = True
            banned[2, logits[2].float().argmax()] = True                            # the best token of a row is banned
        u = torch.rand(Bn, generator=g).clamp_min(1e-6).to(DEV)
        ids = torch.full((Bn, 3), -1, dtype=torch.long, device=DEV)
        o.sample_topk(logits, temp, top_k, u, ids[:, 1], banned)
        z = logits.float() / temp
        if banned is not None:
            z = z.masked_fill(banned[:, :V], float("-inf"))
        z = decoding.batch_top_k_top_p_sampling(z, top_k=top_k, top_p=0.0)
        prob = torch.softmax(z.double(), -1)
        want = decoding.draw_from_uniform(prob.float(), u).view(-1)
        got = ids[:, 1]
        assert (ids[:, 0] == -1).all() and (ids[:, 2] == -1).all()                   # strided output: neighbours untouched
        assert ((got >= 0) & (got < V)).all()
        assert (prob.gather(1, got[:, None]) > 0).all(), case                        # never a filtered / banned token
        c = torch.cumsum(prob, -1)
        for b in (got != want).nonzero().view(-1).tolist():
            lo, hi = sorted((int(got[b]), int(want[b])))
            # a disagreement is legitimate only when u sits on a CDF step: the mass strictly between the two picks is ~0
            assert abs(float(c[b, hi - 1] - c[b, lo]) if hi - 1 >= lo else 0.0) < 1e-5 and \
                min(abs(float(c[b, lo]) - float(u[b])), abs(float(c[b, hi - 1]) - float(u[b]))) < 1e-5, (case, b)
        assert (got != want).sum().item() <= 1, case


def test_sample_topk_degenerate_rows():
    """Rows with a single candidate left, with every token banned, and k larger than the vocabulary: always a valid id, the
    only candidate when there is one."""
    o = ops()
    V, Bn = 300, 4
    logits = rnd(Bn, V, seed=77) * 3
    banned = torch.zeros(Bn, V + 1, dtype=torch.bool, device=DEV)
    banned[0, :V] = True; banned[0, 123] = False                 # one survivor
    banned[1, :V] = True                                         # nothing left
    u = torch.tensor([0.9, 0.5, 1e-7, 0.999999], device=DEV)
    ids = torch.full((Bn,), -1, dtype=torch.long, device=DEV)
    o.sample_topk(logits, 0.7, 7, u, ids, banned)
    assert ids[0].item() == 123 and 0 <= ids[1].item() < V
    o.sample_topk(logits, 1.0, 64, u, ids)                       # k = 64 of 300: rows 2 / 3 draw from the two ends of the CDF
    top = torch.topk(logits.float(), 64, -1).indices
    assert all(int(ids[b]) in set(top[b].tolist()) for b in range(Bn))
    small = logits[:, :40].contiguous()
    o.sample_topk(small, 1.0, 64, u, ids)                        # k beyond the vocabulary = no filter
    assert ((ids >= 0) & (ids < 40)).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_kv_cache_strides_and_shared_kv(dtype):
    """Forward-only descriptor extras: Lq = 1 against a partially filled [B, Umax, H] cache (kv_bstride) and K/V shared
    by groups of batch rows (kv_group)."""
    o = ops()
    Bn, nh, d, Umax, Lk = 5, 4, 32, 19, 7
    H = nh * d
    q = rnd(Bn, H, dtype=dtype, seed=110)
    kc, vc = rnd(Bn * Umax, H, dtype=dtype, seed=111), rnd(Bn * Umax, H, dtype=dtype, seed=112)
    out = torch.empty(Bn, H, device=DEV, dtype=dtype)
    lse = torch.empty(Bn, nh, 1, device=DEV)
    a = o.attn_desc(q, kc, vc, out, lse, None, Bn, nh, 1, Lk, d, kv_bstride=Umax)
    o.attn_fwd(a)
    k3, v3 = kc.view(Bn, Umax, nh, d)[:, :Lk].float(), vc.view(Bn, Umax, nh, d)[:, :Lk].float()
    ref, _ = attn_ref(q.float().view(Bn, 1, nh, d), k3, v3, torch.ones(Bn, Lk, device=DEV), False, -1e4, 1 / math.sqrt(d), None)
    check("attn_cache", out.view(Bn, 1, nh, d), ref, dtype, 2.0)
    # shared K/V: 6 query rows, 2 K/V rows, groups of 3
    Bq, G, Lq, Lk2 = 6, 3, 9, 21
    Q = rnd(Bq * Lq, H, dtype=dtype, seed=113)
    K, V = rnd(2 * Lk2, H, dtype=dtype, seed=114), rnd(2 * Lk2, H, dtype=dtype, seed=115)
    km = torch.ones(2, Lk2, device=DEV)
    km[1, 15:] = 0
    O = torch.empty(Bq * Lq, H, device=DEV, dtype=dtype)
    lse2 = torch.empty(Bq, nh, Lq, device=DEV)
    a = o.attn_desc(Q, K, V, O, lse2, km, Bq, nh, Lq, Lk2, d, mask_neg=-1e9, kv_group=G)
    o.attn_fwd(a)
    rep = lambda t: t.repeat_interleave(G, dim=0)
    ref, _ = attn_ref(Q.float().view(Bq, Lq, nh, d), rep(K.float().view(2, Lk2, nh, d)), rep(V.float().view(2, Lk2, nh, d)),
                      rep(km), False, -1e9, 1 / math.sqrt(d), None)
    check("attn_shared_kv", O.view(Bq, Lq, nh, d), ref, dtype, 2.0)


def test_gemm_randomised_regression():
    """tools/gemm_fuzz.py: random shapes / layouts / leading-dimension padding / epilogue combinations (bias, addend, GELU
    with its aux output, x gelu', dropout, alpha, fp32 output) against torch fp32 -- every tile kernel, the narrow-tile
    variants, split-K and both epilogue paths get hit."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gemm_fuzz", os.path.join(root, "tools", "gemm_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(120, seed=11, verbose=False) == 0


@pytest.mark.parametrize("M,N,K", [(16, 768, 768), (16, 3072, 768), (16, 768, 3072), (3, 2304, 768), (1, 136, 200), (16, 30528, 768),
                                   (7, 64, 40)])
@pytest.mark.parametrize("out", [torch.bfloat16, torch.float32])
def test_skinny_decode_gemm(M, N, K, out):
    """csrc/gemv.hip: the M <= 16 shape of a KV-cached decode step (one new token per row): bias / residual add / GELU (+ gelu'),
    strided output rows (written straight into a cache row), tails in N and K, bf16 and fp32 outputs; and the library really
    dispatches this shape to the skinny kernel."""
    import ctypes as C
    o = ops()
    x, w, b = rnd(M, K, dtype=torch.bfloat16, seed=40), rnd(N, K, dtype=torch.bfloat16, seed=41, s=0.1), rnd(N, seed=42)
    ref = x.float() @ w.float().t() + b
    big = torch.zeros(M, 3, N, device=DEV, dtype=out)               # output rows with a stride of 3N (row t = 1 of a [M, 3, N] cache)
    y = big[:, 1]
    o.gemm(x, w, y, M, N, K, bias=b)
    check("skinny_bias", y.contiguous(), ref, torch.bfloat16)
    assert float(big[:, 0].abs().max()) == 0.0 and float(big[:, 2].abs().max()) == 0.0
    if out == torch.bfloat16:
        r = rnd(M, N, dtype=torch.bfloat16, seed=43)
        y2 = torch.empty(M, N, device=DEV, dtype=out)
        o.gemm(x, w, y2, M, N, K, bias=b, addend=r)
        check("skinny_add", y2, ref + r.float(), torch.bfloat16)
        u = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        a = torch.empty(M, N, device=DEV, dtype=out)
        o.gemm(x, w, a, M, N, K, bias=b, aux=u, epi=o.EPI_GELU)
        uref = ref.clone().requires_grad_(True)
        aref = torch.nn.functional.gelu(uref)
        aref.backward(torch.ones_like(aref))
        check("skinny_gelu", a, aref.detach(), torch.bfloat16)
        check("skinny_gelu_deriv", u, uref.grad, torch.bfloat16)
    else:
        # fp32 output + residual add: the addend has the OUTPUT's type (the skinny kernels read it as bf16)
        r = rnd(M, N, seed=43)
        y2 = torch.empty(M, N, device=DEV, dtype=out)
        o.gemm(x, w, y2, M, N, K, bias=b, addend=r)
        check("skinny_add_f32", y2, ref + r, torch.bfloat16)
        if K <= 1024 and K % 8 == 0:
            gamma, beta = 1.0 + rnd(K, seed=44, s=0.1), rnd(K, seed=45, s=0.1)
            y3 = torch.empty(M, N, device=DEV, dtype=out)
            o.gemv_ln(x, w, y3, M, N, K, gamma, beta, 1e-12, bias=b, addend=r)
            xn = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-12).to(torch.bfloat16).float()
            check("skinny_ln_add_f32", y3, xn @ w.float().t() + b + r, torch.bfloat16)
    from gst_visdial_amd import _lib as L
    d = L.GemmDesc()
    d.A, d.B, d.C, d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.batch = x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, 3 * N, 1
    d.dtype_in, d.dtype_out, d.alpha = L.BF16, (L.BF16 if out == torch.bfloat16 else L.F32), 1.0
    assert "gemv16_kernel" in o.gemm_kernel_symbol(d)


# ---- round 3: LayerNorm folded into the Linear next to it (csrc/gemm_rows.hip), decoder row counts --------------------------
@pytest.mark.parametrize("M,N,H,p,b_km,gelu", [(400, 768, 768, 0.3, False, False), (400, 3072, 768, 0.25, False, True),
                                               (250, 2304, 768, 0.0, False, False), (77, 648, 128, 0.3, False, False),
                                               (400, 768, 768, 0.3, True, False),
                                               # round 5: the vision stream's width (H = 1024: four 256-column vectors per row)
                                               (592, 1024, 1024, 0.1, False, True), (370, 3072, 1024, 0.1, False, False),
                                               (100, 896, 832, 0.2, True, False)])
def test_layernorm_folded_into_the_next_linear_forward(M, N, H, p, b_km, gelu):
    """gstvd_gemm_ln_fwd == gstvd_ln_fwd followed by gstvd_gemm: the normalised rows / mean / rstd it writes are BIT-identical to
    the stand-alone LayerNorm kernel's (same arithmetic in the same order), the product equals the two-launch result (same bf16
    operands, same fp32 accumulation order) and the torch fp32 reference within the bf16 bar; ragged M / N tiles, dropout on the
    LayerNorm's input, GELU (+ saved derivative) epilogue, k-major B."""
    o = ops()
    bf = torch.bfloat16
    x, res = rnd(M, H, dtype=bf, seed=70), rnd(M, H, dtype=bf, seed=71)
    gamma, beta = 1 + 0.1 * rnd(H, seed=72), 0.1 * rnd(H, seed=73)
    w = rnd(N, H, dtype=bf, seed=74, s=0.1)
    wk = w.t().contiguous() if b_km else w
    bias = rnd(N, seed=75)
    rng = o.Rng(DEV, seed=5)

    def mk():
        return dict(mode=o.LN_RESID, dtype=o.BF16, M=M, H=H, gamma=gamma, beta=beta, mean=torch.empty(M, device=DEV),
                    rstd=torch.empty(M, device=DEV), eps=1e-12, x=x, res=res, y=torch.empty(M, H, device=DEV, dtype=bf),
                    p_pre=p, site_pre=9, rng=rng)
    k1, k2 = mk(), mk()
    epi = o.EPI_GELU if gelu else 0
    c1, c2 = torch.zeros(M, N, device=DEV, dtype=bf), torch.zeros(M, N, device=DEV, dtype=bf)
    u1 = torch.zeros(M, N, device=DEV, dtype=bf) if gelu else None
    u2 = torch.zeros(M, N, device=DEV, dtype=bf) if gelu else None
    o.ln_fwd(**k1)
    o.gemm(k1["y"], wk, c1, M, N, H, b_km=b_km, bias=bias, aux=u1, epi=epi)
    # (gemm_ln_ok is the ENGINE's policy -- shapes whose grid is a single round of the chip; the kernel itself takes all of these)
    assert o.gemm_ln_ok(M, N, H, bf) == (((M + 15) // 16) * ((N + 127) // 128) <= 512 and H <= o.LN_FOLD_MAX_H)
    o.gemm_ln_fwd(k2, wk, c2, N, b_km=b_km, bias=bias, aux=u2, epi=epi)
    assert torch.equal(k1["y"], k2["y"]) and torch.equal(k1["mean"], k2["mean"])
    # (H = 768, the decoder's width, runs the same three-vector code as ln_fwd's instantiation: bit-equal; narrower rows take
    # another instantiation there, whose fused-multiply-add contraction may differ in the last bit of the variance)
    assert torch.equal(k1["rstd"], k2["rstd"]) if H in (768, 1024) else torch.allclose(k1["rstd"], k2["rstd"], rtol=1e-6, atol=0)
    assert torch.equal(c1, c2)
    if gelu:
        assert torch.equal(u1, u2)
    mask = o.dropout_mask(M * H, p, 9, rng, DEV).view(M, H) if p > 0 else torch.ones(M, H, device=DEV)
    yref = ln_ref(x.float() * mask + res.float(), gamma, beta).to(bf).float()
    ref = yref @ w.float().t() + bias
    check("gemm_ln_fwd", c2, torch.nn.functional.gelu(ref) if gelu else ref, bf, 2.0)


@pytest.mark.parametrize("M,N,H,p,dgelu,add", [(400, 768, 768, 0.3, False, True), (400, 3072, 768, 0.25, True, False),
                                               (250, 768, 768, 0.0, False, False), (77, 648, 128, 0.3, False, True),
                                               (592, 1024, 1024, 0.1, True, False), (592, 1024, 1024, 0.1, False, True),
                                               (100, 896, 832, 0.2, False, False)])
def test_layernorm_backward_folded_into_the_producers_input_gradient(M, N, H, p, dgelu, add):
    """gstvd_gemm_ln_bwd == gstvd_ln_bwd followed by the input-gradient GEMM of the Linear that produced the LayerNorm's input
    (C = dx . W, W row-major [H, N] = k-major B): dres / dx bit-identical to the stand-alone kernel, the product equal to the
    two-launch result, the column partials (16-row blocks of two rows per wave instead of 4-row blocks) equal after reduction up to fp32
    reassociation; residual-gradient accumulate and the x gelu' epilogue of the FFN's input gradient."""
    o = ops()
    bf = torch.bfloat16
    x, res = rnd(M, H, dtype=bf, seed=80), rnd(M, H, dtype=bf, seed=81)
    gamma, beta = 1 + 0.1 * rnd(H, seed=82), 0.1 * rnd(H, seed=83)
    dy = rnd(M, H, dtype=bf, seed=84)
    w = rnd(H, N, dtype=bf, seed=85, s=0.1)                    # the producer Linear's weight [out = H, in = N]
    rng = o.Rng(DEV, seed=6)
    kw = dict(mode=o.LN_RESID, dtype=o.BF16, M=M, H=H, gamma=gamma, beta=beta, mean=torch.empty(M, device=DEV),
              rstd=torch.empty(M, device=DEV), eps=1e-12, x=x, res=res, y=torch.empty(M, H, device=DEV, dtype=bf),
              p_pre=p, site_pre=4, rng=rng)
    o.ln_fwd(**kw)
    addend = rnd(M, N, dtype=bf, seed=86) if add else None
    aux = (rnd(M, N, dtype=bf, seed=87) * 0.5 + 0.5) if dgelu else None
    epi = o.EPI_DGELU if dgelu else 0
    # two launches
    nb1 = o.ln_bwd_blocks(M)
    p1 = torch.empty(nb1, 3, H, device=DEV)
    dres1 = torch.empty(M, H, device=DEV, dtype=bf)
    dx1 = torch.empty(M, H, device=DEV, dtype=bf) if p > 0 else dres1
    o.ln_bwd(kw, dy, p1, dres=dres1, dx=dx1)
    c1 = torch.zeros(M, N, device=DEV, dtype=bf)
    o.gemm(dx1, w, c1, M, N, H, b_km=True, addend=addend, aux=aux, epi=epi)
    # one launch
    R = o.gemm_ln_rows()
    nb2 = (M + R - 1) // R
    p2 = torch.empty(nb2, 3, H, device=DEV)
    dres2 = torch.empty(M, H, device=DEV, dtype=bf)
    dx2 = torch.empty(M, H, device=DEV, dtype=bf) if p > 0 else dres2
    c2 = torch.zeros(M, N, device=DEV, dtype=bf)
    o.gemm_ln_bwd(kw, dy, p2, nb2, w, c2, N, dres=dres2, dx=dx2, addend=addend, aux=aux, epi=epi)
    assert torch.equal(dres1, dres2) and torch.equal(dx1, dx2)
    assert torch.equal(c1, c2)
    outs = []
    for part, nb in ((p1, nb1), (p2, nb2)):
        dg, db, dbias = torch.empty(H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV)
        o.colsum_partials(part, nb, 3, H, dg, db, dbias, accumulate=False)
        outs.append((dg, db, dbias))
    for a, b, name in zip(outs[0], outs[1], ("dgamma", "dbeta", "dbias")):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item()), name
