#!/usr/bin/env python3
"""Randomised regression of gstvd_gemm (bf16) against torch fp32: random shapes / layouts / epilogues / strides.
usage: gemm_fuzz.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops as o
dev = "cuda"


def run(n_cases=200, seed=0, verbose=True):
    rnd = random.Random(seed)
    bad = 0
    rng = o.Rng(torch.device(dev), seed=5)
    for case in range(n_cases):
        big = rnd.random() < 0.4
        M = rnd.choice([1, 7, 16, 37, 64, 100, 255, 256, 400, 592, 1000, 2048, 2816, 4096]) if not big else rnd.choice([2048, 2816, 3000, 4096, 4688])
        N = rnd.choice([8, 64, 96, 136, 192, 256, 760, 768, 1024, 1160, 2304, 3072]) if not big else rnd.choice([768, 1024, 1536, 2304, 2824, 3072])
        K = rnd.choice([8, 64, 72, 200, 768, 1024, 1992, 3072])
        lay = rnd.choice(["nt", "nn", "tn"])
        a_km, b_km = lay == "tn", lay in ("nn", "tn")
        if a_km: M = (M + 7) // 8 * 8
        if b_km: N = (N + 7) // 8 * 8
        pad = rnd.choice([0, 8, 24])                                       # leading-dimension padding
        bf = torch.bfloat16
        g = torch.Generator(device="cpu").manual_seed(case)
        def mk(r, c):
            t = (torch.randn(r, c + pad, generator=g) * 0.5).to(dev).to(bf)
            return t[:, :c]
        A = mk(K, M) if a_km else mk(M, K)
        B = mk(K, N) if b_km else mk(N, K)
        out_f32 = rnd.random() < 0.15
        C = torch.empty(M, N + pad, device=dev, dtype=torch.float32 if out_f32 else bf)[:, :N]
        kw, ref = {}, (A.float().t() if a_km else A.float()) @ (B.float() if b_km else B.float().t())
        alpha = rnd.choice([1.0, 1.0, 0.125])
        ref = ref * alpha
        if rnd.random() < 0.6:
            bias = torch.randn(N, generator=g).to(dev); kw["bias"] = bias; ref = ref + bias
        if rnd.random() < 0.4:
            add = mk(M, N) if not out_f32 else torch.randn(M, N, generator=g).to(dev); kw["addend"] = add; ref = ref + add.float()
        epi, aux = 0, None
        mode = rnd.choice(["none", "none", "gelu", "dgelu"]) if not out_f32 else "none"
        if mode == "gelu":
            aux = torch.empty(M, N + pad, device=dev, dtype=bf)[:, :N]; epi = o.EPI_GELU
            u = ref.clone().requires_grad_(True); r2 = torch.nn.functional.gelu(u); r2.backward(torch.ones_like(r2)); aux_ref = u.grad; ref = r2.detach()
        elif mode == "dgelu":
            aux = mk(M, N); epi = o.EPI_DGELU; ref = ref * aux.float()
        drop = rnd.random() < 0.25 and not out_f32
        if drop:
            kw.update(drop_p=0.3, site=7 + case, rng=rng)
        o.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km, aux=aux, epi=epi, alpha=alpha, **kw)
        if drop:
            mask = o.dropout_mask(M * N, 0.3, 7 + case, rng, torch.device(dev))
            ref = ref * mask.view(M, N)
        torch.cuda.synchronize()
        scale = max(ref.abs().max().item(), 1e-6)
        err = (C.float() - ref).abs().max().item() / scale
        ok = err <= (2.5e-2 if not out_f32 else 1e-2) and torch.isfinite(C.float()).all()
        if mode == "gelu":
            e2 = (aux.float() - aux_ref).abs().max().item()
            ok = ok and e2 <= 2e-2
        if not ok:
            bad += 1
            print("FAIL case %d: %s M=%d N=%d K=%d pad=%d f32out=%s mode=%s drop=%s alpha=%g kw=%s err=%.3e" % (case, lay, M, N, K, pad, out_f32, mode, drop, alpha, sorted(kw), err))
    if verbose:
        print("%d cases, %d failures" % (n_cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
