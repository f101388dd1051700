#!/usr/bin/env python3
"""How much longer does a kernel of the replayed step take when a kernel of ANOTHER hardware queue is in flight?  Reads a compressed
rocprofv3 kernel trace of replayed steps (tools/trace_step.sh -> gpurun_out/trace/kernel_trace.csv.gz; the committed one is
profiles/r04_kernel_trace_steps.csv.gz), takes one steady-state step (rng_advance -> next rng_advance) and prints, per kernel
family, launches and mean duration alone / overlapped.   usage: overlap_stats.py [trace.csv.gz] [step index, default 5]   (no GPU needed)"""
import csv, gzip, io, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_kernel_trace_steps.csv.gz")
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = sorted(csv.DictReader(io.TextIOWrapper(gzip.open(path))), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rng_advance" in r["Kernel_Name"]]
st = rows[idx[which]:idx[which + 1]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in st]
FAM = [("ln_bwd 16 waves (4096 rows)", "ln_bwd_kernelIDF16bLi0ELi3ELi1ELi16"), ("ln_bwd 4 waves", "ln_bwd_kernelIDF16bLi0ELi4ELi1ELi4"),
       ("ln_fwd H<=768", "ln_fwd_kernelIDF16bLi0ELi3E"), ("ln_fwd H=1024 (vision)", "ln_fwd_kernelIDF16bLi0ELi4E"),
       ("gemm 128-tile NN", "gemm_dma_kernelIDF16bLi128ELi128ELi4ELi2ELb0ELb1"), ("gemm 128-tile NT", "gemm_dma_kernelIDF16bLi128ELi128ELi4ELi2ELb0ELb0"),
       ("gemm 256-tile NT (full-line)", "gemm_pc256_nt64"), ("gemm 256-tile NN", "gemm_pc256_kernelIDF16bLb0ELb1"),
       ("gemm 64-tile NN", "gemm_dma_kernelIDF16bLi64ELi64ELi2ELi2ELb0ELb1"), ("gemm 64-tile NT", "gemm_dma_kernelIDF16bLi64ELi64ELi2ELi2ELb0ELb0"),
       ("attn_fwd d=64", "attn_fwd_kernelIDF16bLi64"), ("attn_bwd d=64", "attn_bwd_kernelIDF16bLi64"),
       ("attn_fwd d=128", "attn_fwd_kernelIDF16bLi128"), ("attn_bwd d=128", "attn_bwd_kernelIDF16bLi128"),
       ("attn_bwd one pass (round 5)", "attn_bwd_onepass")]
print("step %d of %s: %d kernels, %.3f ms" % (which, os.path.basename(path), len(ev), (max(e for _, e, _, _ in ev) - ev[0][0]) / 1e6))
print("%-32s %8s %10s %8s %12s" % ("kernel family", "alone", "mean us", "overl.", "mean us"))
for label, key in FAM:
    alone, over = [], []
    for s, e, n, q in ev:
        if key in n:
            hit = any(q2 != q and s2 < e and e2 > s for s2, e2, n2, q2 in ev)
            (over if hit else alone).append((e - s) / 1e3)
    if alone or over:
        print("%-32s %8d %10s %8d %12s" % (label, len(alone), "%.1f" % statistics.mean(alone) if alone else "-", len(over),
                                           "%.1f" % statistics.mean(over) if over else "-"))
