#!/usr/bin/env python3
"""K-slope of the vendor BLAS (torch.matmul) vs this library's 256-tile NT kernel at M x N = 4096 x 3072 and 4096 x 2304:
t = fixed + slope * (K / 32) -- where does the 20-27 % gap of profiles/r03_gemm_vs_vendor_blas.txt sit, in the K loop or in
the fixed cost (ring fill + epilogue)?   usage: nt_study.py   (GSTVD_GEMM_* tuning variables apply to this library's side)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
dev = "cuda"


def t_us(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def fit(res):
    n = len(res); sx = sum(r[0] for r in res); sy = sum(r[1] for r in res)
    sxx = sum(r[0] ** 2 for r in res); sxy = sum(r[0] * r[1] for r in res)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    return (sy - slope * sx) / n, slope


for M, N in ((4096, 3072), (4096, 2304), (4096, 768)):
    ours, lib = [], []
    for K in (384, 768, 1536, 3072):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ours.append((K // 32, t_us(lambda: ops.gemm(A, B, C, M, N, K))))
        Bt = B.t()
        lib.append((K // 32, t_us(lambda: torch.matmul(A, Bt, out=C))))
    fo, so = fit(ours); fl, sl = fit(lib)
    print("nt %dx%d  ours: %s -> fixed %.1f us + %.3f us/step   vendor: %s -> fixed %.1f us + %.3f us/step"
          % (M, N, " ".join("%d:%.1f" % r for r in ours), fo, so, " ".join("%d:%.1f" % r for r in lib), fl, sl))
