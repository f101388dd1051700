#!/usr/bin/env python3
"""Run one GEMM shape through the C ABI repeatedly (for rocprofv3 counters / timing).
usage: gemm_probe.py LAYOUT M N K [reps]   LAYOUT in nt|nn|tn ; bf16 in, bf16 out (tn: fp32 out)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops

lay, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda"
a_km, b_km = (lay == "tn"), (lay in ("nn", "tn"))
A = torch.randn((K, M) if a_km else (M, K), device=dev).to(torch.bfloat16)
B = torch.randn((K, N) if b_km else (N, K), device=dev).to(torch.bfloat16)
C = torch.empty(M, N, device=dev, dtype=torch.float32 if lay == "tn" else torch.bfloat16)
for _ in range(3):
    ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("%s %dx%dx%d  %.1f us  %.1f TFLOP/s" % (lay, M, N, K, us, 2.0 * M * N * K / us / 1e6))
