#!/usr/bin/env python3
"""How much of an in-step GEMM's time is cold operands?  The step's large GEMMs run 20-28 % slower inside the step (serialized,
one stream) than back to back in a micro-benchmark; there the 4.7 MB weight matrix and the activations stay in the 256 MB
Infinity Cache, in the step 0.8 GB of weights stream through it between two uses.  Per shape, HIP-event time of ONE launch:
  hot        the same operands again (the micro-benchmark's situation)
  cold       after a 1 GiB fill of another buffer (nothing of A / B / C left in L2 / Infinity Cache)
  cold+B     cold, then the WEIGHTS touched by a streaming read (what a prefetch launch on a side stream would do)
  cold+A     cold, then the ACTIVATIONS touched (what the producing kernel leaves behind in the step)
  cold+AB    both"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
dev = "cuda"
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)          # 1 GiB

def once(fn, prep):
    ts = []
    for _ in range(7):
        prep()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for lay, M, N, K in [("nt", 4096, 3072, 768), ("nn", 4096, 3072, 768), ("nt", 4096, 768, 3072), ("nn", 4096, 768, 3072), ("nt", 4096, 768, 768),
                     ("nt", 592, 1024, 1024), ("nt", 400, 768, 768), ("nt", 400, 3072, 768)]:
    a_km, b_km = False, lay == "nn"
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if b_km else (N, K), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(A, B, C, M, N, K, a_km=a_km, b_km=b_km)
    for _ in range(3): fn()
    cold = lambda: flush.fill_(1.0)
    def touch(*ts):
        def f():
            flush.fill_(1.0)
            for t in ts: t.float().sum()
        return f
    res = dict(hot=once(fn, lambda: None), cold=once(fn, cold), coldB=once(fn, touch(B)), coldA=once(fn, touch(A)), coldAB=once(fn, touch(A, B)))
    print("%s %5dx%5dx%5d  hot %5.1f  cold %5.1f  cold+B %5.1f  cold+A %5.1f  cold+AB %5.1f us" % (lay, M, N, K, res["hot"], res["cold"], res["coldB"], res["coldA"], res["coldAB"]))
