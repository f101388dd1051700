#!/usr/bin/env python3
"""Back-to-back attention forward / backward at one of the step's shapes (for rocprofv3 --pmc / timing).
usage: attn_probe.py B nh Lq Lk d [causal] [p_drop] [reps] [bits]     (bits = 1: forward leaves its dropout keep bits for backward, as in the step)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
B, nh, Lq, Lk, d = [int(x) for x in sys.argv[1:6]]
causal = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
p = float(sys.argv[7]) if len(sys.argv) > 7 else 0.1
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev, bf = "cuda", torch.bfloat16
H = nh * d
Q = torch.randn(B * Lq, H, device=dev).to(bf); K = torch.randn(B * Lk, H, device=dev).to(bf); V = torch.randn(B * Lk, H, device=dev).to(bf)
O = torch.empty(B * Lq, H, device=dev, dtype=bf); dO = torch.randn(B * Lq, H, device=dev).to(bf)
dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
lse = torch.empty(B * nh * Lq, device=dev); delta = torch.empty_like(lse)
mask = torch.ones(B, Lk, device=dev); mask[:, int(0.8 * Lk):] = 0
rng = ops.Rng(torch.device(dev), seed=1)
use_bits = len(sys.argv) > 9 and int(sys.argv[9]) != 0
nbits = ops.attn_keep_bits_shape(B, nh, Lq, Lk, d, torch.bfloat16, causal, p) if use_bits else 0
bits = torch.zeros(nbits, device=dev, dtype=torch.int64) if nbits else None
a = ops.attn_desc(Q, K, V, O, lse, mask, B, nh, Lq, Lk, d, causal=causal, mask_neg=-10000.0, drop_p=p, site=5, rng=rng, drop_bits=bits)
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
tf = t(lambda: ops.attn_fwd(a))
tb = t(lambda: ops.attn_bwd(a, dO, dQ, dK, dV, delta))
fl = 4.0 * B * nh * Lq * Lk * d
print("attn B=%d nh=%d Lq=%d Lk=%d d=%d causal=%d p=%.2f bits=%d: fwd %.1f us (%.0f TF/s)  bwd %.1f us (%.0f TF/s)" % (B, nh, Lq, Lk, d, causal, p, int(bits is not None), tf, fl / tf / 1e6, tb, 3.5 * fl / tb / 1e6))
