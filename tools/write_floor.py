#!/usr/bin/env python3
"""How fast can this GPU write an output of GEMM size (the epilogue's floor)?  torch fill / copy of the same byte counts."""
import torch
for mb, name in ((25.2, "4096x3072 bf16"), (33.6, "4096x4096 bf16"), (6.3, "4096x768 bf16"), (172.8, "4688x18432 bf16")):
    n = int(mb * 1e6 / 2)
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
    for op, f in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: x.copy_(y))):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("%-18s %s: %6.1f us  (%.2f TB/s written)" % (name, op, us, mb * 1e6 / us / 1e6))
