#!/usr/bin/env python3
"""Stand-alone timing of the fused sampling kernel (gstvd_sample_topk): 16 rows x 30522 logits, by top_k and dtype."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
dev = "cuda"
for dtype in (torch.float32, torch.bfloat16):
    x = (torch.randn(16, 30528, device=dev) * 2.5).to(dtype)[:, :30522]
    u = torch.rand(16, device=dev).clamp_min(1e-6)
    out = torch.zeros(16, dtype=torch.long, device=dev)
    for k in (0, 1, 7, 64):
        for _ in range(3): ops.sample_topk(x, 0.7, k, u, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.sample_topk(x, 0.7, k, u, out)
        e1.record(); torch.cuda.synchronize()
        print("%s top_k=%2d: %.1f us" % (str(dtype)[6:], k, e0.elapsed_time(e1) * 20))
