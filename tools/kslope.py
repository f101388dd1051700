#!/usr/bin/env python3
"""Per-K-step time and fixed cost of a GEMM tile class: times LAYOUT MxNxK at several K and fits t = fixed + slope * steps.
usage: kslope.py LAYOUT M N [kstep=32]   (tuning env vars GSTVD_GEMM_* apply; with GSTVD_DIAG_ABLATE set the DIAGNOSTIC library
is built and loaded instead of the product one -- tools/diag_lib.py)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("GSTVD_DIAG_ABLATE"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import diag_lib
    diag_lib.use()
import torch
from gst_visdial_amd import ops
lay, M, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kstep = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = "cuda"
a_km, b_km = (lay == "tn"), (lay in ("nn", "tn"))
res = []
for K in (768, 1536, 3072, 6144):
    A = torch.randn((K, M) if a_km else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if b_km else (N, K), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if lay == "tn" else torch.bfloat16)
    for _ in range(3): ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    e1.record(); torch.cuda.synchronize()
    res.append((K // kstep, e0.elapsed_time(e1) * 1e3 / 20))
n = len(res); sx = sum(r[0] for r in res); sy = sum(r[1] for r in res)
sxx = sum(r[0] ** 2 for r in res); sxy = sum(r[0] * r[1] for r in res)
slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); fixed = (sy - slope * sx) / n
print("%s %dx%d %s: %s  -> fixed %.1f us + %.3f us/step" % (lay, M, N, ops.gemm_tag(1, a_km, b_km, M, N, 1),
      " ".join("%d:%.1f" % r for r in res), fixed, slope))
