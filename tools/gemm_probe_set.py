#!/usr/bin/env python3
"""Back-to-back timing of the step's main GEMM shapes (bf16)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
dev = "cuda"
def run(lay, M, N, K, reps=30):
    a_km, b_km = (lay == "tn"), (lay in ("nn", "tn"))
    A = torch.randn((K, M) if a_km else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if b_km else (N, K), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if lay == "tn" else torch.bfloat16)
    for _ in range(3): ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%s %5dx%5dx%5d %-16s %7.1f us  %6.1f TFLOP/s" % (lay, M, N, K, ops.gemm_tag(1, a_km, b_km, M, N, 1), us, 2.0 * M * N * K / us / 1e6))
for sh in [("nt", 4096, 3072, 768), ("nt", 4096, 2304, 768), ("nn", 4096, 3072, 768), ("nt", 4688, 18432, 768), ("nn", 4688, 768, 18432),
           ("tn", 18432, 768, 4688), ("tn", 3072, 768, 4096), ("nt", 4096, 768, 3072), ("nn", 4096, 768, 3072), ("nt", 4096, 768, 768),
           ("nt", 592, 1024, 1024), ("nn", 592, 1024, 1024), ("nt", 400, 768, 768), ("nn", 400, 768, 3072), ("nt", 400, 30528, 768)]:
    run(*sh)
