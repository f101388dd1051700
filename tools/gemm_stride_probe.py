#!/usr/bin/env python3
"""Does the row stride of the operands matter? Same GEMM, leading dimensions padded by `pad` elements."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops

def run(lay, M, N, K, pa, pb, pc, reps=30):
    dev = "cuda"
    a_km, b_km = (lay == "tn"), (lay in ("nn", "tn"))
    ra, ca = ((K, M) if a_km else (M, K))
    rb, cb = ((K, N) if b_km else (N, K))
    A = torch.randn(ra, ca + pa, device=dev).to(torch.bfloat16)[:, :ca]
    B = torch.randn(rb, cb + pb, device=dev).to(torch.bfloat16)[:, :cb]
    C = torch.empty(M, N + pc, device=dev, dtype=torch.float32 if lay == "tn" else torch.bfloat16)[:, :N]
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%s %5dx%5dx%5d pad(a,b,c)=(%3d,%3d,%3d)  %7.1f us  %6.1f TFLOP/s" % (lay, M, N, K, pa, pb, pc, us, 2.0 * M * N * K / us / 1e6))

for lay, M, N, K in [("nt", 4096, 3072, 768), ("nn", 4096, 768, 3072), ("nn", 4096, 3072, 768), ("tn", 3072, 768, 4096), ("tn", 768, 3072, 4096), ("nt", 4096, 768, 3072)]:
    for pads in [(0, 0, 0), (64, 64, 64), (8, 8, 8), (136, 136, 136)]:
        run(lay, M, N, K, *pads)
