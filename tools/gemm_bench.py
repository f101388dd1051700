#!/usr/bin/env python3
"""Back-to-back timing + correctness of the step's main GEMM shapes through the C ABI (bf16 in/out; tn: fp32 out).
usage: gemm_bench.py [set] [lib]   set in: main (default) | small | mscale | all.  Tuning env vars (GSTVD_GEMM_*) apply.
`lib`: also time torch.matmul (the vendor BLAS behind it) on the same operands -- a speed-of-light REFERENCE for the tile design,
never a code path of the product (which has no BLAS call)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
dev = "cuda"
MAIN = [("nt", 4096, 3072, 768), ("nn", 4096, 3072, 768), ("nt", 4096, 2304, 768), ("nt", 4096, 768, 3072), ("nn", 4096, 768, 3072),
        ("nn", 4096, 768, 2304), ("nt", 4096, 768, 768), ("nn", 4096, 768, 768), ("nt", 4688, 18432, 768), ("nn", 4688, 768, 18432),
        ("tn", 18432, 768, 4688), ("tn", 3072, 768, 4096), ("tn", 768, 3072, 4096)]
SMALL = [("nt", 592, 1024, 1024), ("nn", 592, 1024, 1024), ("nt", 592, 3072, 1024), ("nn", 592, 1024, 3072), ("nt", 400, 768, 768),
         ("nn", 400, 768, 768), ("nt", 400, 2304, 768), ("nt", 400, 3072, 768), ("nn", 400, 768, 3072), ("nt", 400, 768, 3072),
         ("nt", 400, 30528, 768), ("nn", 400, 768, 30528)]
def run(lay, M, N, K, reps=30):
    a_km, b_km = (lay == "tn"), (lay in ("nn", "tn"))
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((K, M) if a_km else (M, K), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn((K, N) if b_km else (N, K), device=dev, generator=g).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if lay == "tn" else torch.bfloat16)
    for _ in range(3): ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (64,), device=dev)
    Af = (A.t() if a_km else A).float()[rows]
    ref = Af @ (B.float() if b_km else B.float().t())
    err = ((C[rows].float() - ref).abs().max() / ref.abs().max()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    lib = ""
    if len(sys.argv) > 2 and sys.argv[2] == "lib":
        Am = A.t() if a_km else A
        Bm = B if b_km else B.t()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3): torch.matmul(Am, Bm, out=out)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps): torch.matmul(Am, Bm, out=out)
        e1.record(); torch.cuda.synchronize()
        ul = e0.elapsed_time(e1) * 1e3 / reps
        lib = "   | vendor BLAS (torch.matmul, bf16 out) %7.1f us %6.1f TFLOP/s" % (ul, 2.0 * M * N * K / ul / 1e6)
    print("%s %5dx%5dx%5d %-22s %7.1f us  %6.1f TFLOP/s  relerr %.1e%s%s" % (lay, M, N, K, ops.gemm_tag(1, a_km, b_km, M, N, 1), us,
          2.0 * M * N * K / us / 1e6, err, "  <-- WRONG" if err > 2e-2 else "", lib))
# row scaling of the latency-bound decoder / vision problems: would running the decoder's rows as two half-size chains on two
# streams shorten it?  Only if a launch gets faster with fewer rows.
MSCALE = [(lay, M, N, K) for (lay, N, K) in (("nt", 768, 768), ("nn", 768, 768), ("nt", 2304, 768), ("nt", 3072, 768), ("nn", 768, 3072))
          for M in (400, 200, 100)]
which = sys.argv[1] if len(sys.argv) > 1 else "main"
for sh in (MAIN if which == "main" else SMALL if which == "small" else MSCALE if which == "mscale" else MAIN + SMALL):
    run(*sh)
