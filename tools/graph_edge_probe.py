#!/usr/bin/env python3
"""What a cross-stream dependency costs the SOURCE chain inside a replayed hipGraph.

profiles/r06_timeline.txt: the 0.2 ms per step in which no kernel runs is 24 intervals of 9-11 us, each at a point where a kernel of
the text chain has a dependant on the other stream (co-attention: the vision side reads the text side's K / V; backward: the reverse) --
and the text chain's OWN next kernel starts late by the same amount.  This probe isolates it: a chain of CHAIN dependent kernels on
stream A, captured in one graph, with
  none     no other stream
  fork     after every chain kernel a small kernel on stream B waits for it (A -> B edges; B joins A at the end only)
  join     before every chain kernel, A waits for a small independent kernel of stream B (B -> A edges)
  both     fork + join at every kernel (the co-attention layer's pattern)
and reports the replayed time per chain kernel.  usage: graph_edge_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gst_visdial_amd import ops
from gst_visdial_amd._lib import LN_RESID

dev = torch.device("cuda")
bf = torch.bfloat16
rng = ops.Rng(dev, seed=1)
CHAIN, REPLAYS = 24, 10


def ln_call(M, H=768):
    x, res, y = [torch.randn(M, H, device=dev).to(bf) for _ in range(3)]
    kw = dict(mode=LN_RESID, dtype=ops.BF16, M=M, H=H, gamma=torch.ones(H, device=dev), beta=torch.zeros(H, device=dev),
              mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev), eps=1e-12, x=x, res=res, y=y, p_pre=0.0, site_pre=3, rng=rng)
    return lambda: ops.ln_fwd(**kw)


def run(mode, every=1):
    big = [ln_call(4096) for _ in range(2)]
    small = [ln_call(64) for _ in range(CHAIN)]
    sB = torch.cuda.Stream()

    def body():
        sA = torch.cuda.current_stream()
        sB.wait_stream(sA)
        for i in range(CHAIN):
            edge = (i % every == 0)
            if edge and mode in ("join", "both"):
                with torch.cuda.stream(sB):
                    small[i]()
                sA.wait_stream(sB)
            big[i & 1]()
            if edge and mode in ("fork", "both"):
                sB.wait_stream(sA)
                with torch.cuda.stream(sB):
                    small[i]()
        sA.wait_stream(sB)

    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
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
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPLAYS)
    return best


if __name__ == "__main__":
    base = run("none")
    print("chain of %d dependent ln_fwd(4096 x 768) launches in one replayed graph: %.1f us = %.2f us per kernel" % (CHAIN, base, base / CHAIN))
    for mode in ("fork", "join", "both"):
        for every in (1, 4):
            t = run(mode, every)
            n = (CHAIN + every - 1) // every
            print("  %-5s at every %d%s kernel (%2d edges each way): %.1f us  -> +%.2f us per edge on the chain" % (
                mode, every, "st" if every == 1 else "th", n, t, (t - base) / n))
