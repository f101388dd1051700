#!/usr/bin/env python3
"""An UPPER BOUND on what deferred normalisation can take off the 4096-row chain, priced with kernels that exist.

Today, per LayerNorm site of the text chain:   producer GEMM (g = A W^T + b)  ->  ln_fwd (y = LN(drop(g) + res))
                                               ->  consumer GEMM (reads y).
Deferred form:  the producer's epilogue writes z = drop(acc + b) + y_prev itself (EPI_BIAS | EPI_DROPOUT | EPI_ADD exist) and the consumer
reads z.  What this probe leaves OUT of the deferred form, all of which costs time: recomputing y_prev from (z_prev, mean, rstd, gamma, beta)
in the producer's epilogue, the per-tile row-statistic partials, `rstd (acc - mean s[n]) + c[n]` in the consumer's epilogue, and in backward the
materialisation of y for the weight-gradient launch.  So (today - deferred) here is the most a site can gain.

Each sequence is captured REP times into one hipGraph and replayed (no host in the timing; dependent kernels of a replayed graph start back
to back).  usage: ln_deferred_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
from gst_visdial_amd._lib import LN_RESID, EPI_GELU

dev = torch.device("cuda")
bf = torch.bfloat16
rng = ops.Rng(dev, seed=1)
REP, REPLAYS = 20, 10


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPLAYS):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (REP * REPLAYS))
    return best


def site(name, M, H, Kp, Nc, gelu, p):
    """producer: [M, Kp] x [H, Kp]^T -> [M, H];  LayerNorm over H;  consumer: [M, H] x [Nc, H]^T -> [M, Nc] (+ GELU)."""
    a = torch.randn(M, Kp, device=dev).to(bf)
    wp = (torch.randn(H, Kp, device=dev) * 0.03).to(bf)
    wc = (torch.randn(Nc, H, device=dev) * 0.03).to(bf)
    bp, bc = torch.randn(H, device=dev) * 0.1, torch.randn(Nc, device=dev) * 0.1
    res = torch.randn(M, H, device=dev).to(bf)
    g, y, z = torch.empty(M, H, device=dev, dtype=bf), torch.empty(M, H, device=dev, dtype=bf), torch.empty(M, H, device=dev, dtype=bf)
    out = torch.empty(M, Nc, device=dev, dtype=bf)
    aux = torch.empty(M, Nc, device=dev, dtype=bf) if gelu else None
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    kw = dict(mode=LN_RESID, dtype=ops.BF16, M=M, H=H, gamma=gamma, beta=beta, mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev),
              eps=1e-12, x=g, res=res, y=y, p_pre=p, site_pre=3, rng=rng)
    epi = EPI_GELU if gelu else 0

    def producer_plain():
        ops.gemm(a, wp, g, M, H, Kp, bias=bp)

    def producer_z():
        ops.gemm(a, wp, z, M, H, Kp, bias=bp, addend=res, drop_p=p, site=3, rng=rng)

    def ln():
        ops.ln_fwd(**kw)

    def consumer(x):
        ops.gemm(x, wc, out, M, Nc, H, bias=bc, aux=aux, epi=epi)

    def today():
        producer_plain(); ln(); consumer(y)

    def deferred():
        producer_z(); consumer(z)

    tp, tz, tl, tc = timed(producer_plain), timed(producer_z), timed(ln), timed(lambda: consumer(y))
    tt, td = timed(today), timed(deferred)
    print("%-28s producer %5.1f  producer writing z %5.1f (+%4.1f)  ln_fwd %5.1f  consumer %5.1f | chain today %6.1f  deferred (upper bound form) %6.1f"
          "  -> at most %4.1f us per site" % (name, tp, tz, tz - tp, tl, tc, tt, td, tt - td))
    return tt - td


if __name__ == "__main__":
    print("us per launch / per chain, hipGraph replay of %d captured repetitions, best of 3; p = dropout probability of the site" % REP)
    tot = 0.0
    for p in (0.1, 0.0):
        d1 = site("ln1: ao -> LN -> FFN-up  p=%.1f" % p, 4096, 768, 768, 3072, True, p)
        d2 = site("ln2: fo -> LN -> QKV     p=%.1f" % p, 4096, 768, 3072, 2304, False, p)
        if p > 0:
            tot = 18 * (d1 + d2)
    print("text chain, training: 18 sites of each kind -> at most %.3f ms per step before the costs the probe leaves out" % (tot / 1e3))
