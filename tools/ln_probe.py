#!/usr/bin/env python3
"""Back-to-back timing of the fused dropout+residual+LayerNorm kernels (bf16) at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
from gst_visdial_amd._lib import LN_RESID
dev = "cuda"
rng = ops.Rng(torch.device(dev), seed=1)
def run(M, H, p, reps=40):
    bf = torch.bfloat16
    x, res, y, dy = [torch.randn(M, H, device=dev).to(bf) for _ in range(4)]
    dres, dx = torch.empty_like(x), torch.empty_like(x)
    g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    kw = dict(mode=LN_RESID, dtype=ops.BF16, M=M, H=H, gamma=g, beta=b, mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev),
              eps=1e-12, x=x, res=res, y=y, p_pre=p, site_pre=3, rng=rng)
    nblk = ops.ln_bwd_blocks(M)
    partial = torch.empty(nblk * 3 * H, device=dev)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    tf = t(lambda: ops.ln_fwd(**kw))
    tb = t(lambda: ops.ln_bwd(kw, dy, partial, dres=dres, dx=dx))
    byt_f, byt_b = 3 * M * H * 2, (5 * M * H * 2 + nblk * 3 * H * 4)
    print("M=%5d H=%4d p=%.1f  fwd %6.1f us (%5.0f GB/s)   bwd %6.1f us (%5.0f GB/s)" % (M, H, p, tf, byt_f / tf / 1e3, tb, byt_b / tb / 1e3))
shapes = ((4096, 768), (592, 1024), (400, 768), (16384, 768))
if len(sys.argv) > 1 and sys.argv[1] == "mscale":      # row scaling of the decoder's LayerNorms (see tools/row_split_probe.sh)
    shapes = ((400, 768), (200, 768), (100, 768))
for M, H in shapes:
    for p in (0.0, 0.3):
        run(M, H, p)
