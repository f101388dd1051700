#!/usr/bin/env python3
"""Premise check for row-split software pipelining: does the GPU run two staggered HALF-batch train steps (8 rows each, own
streams) faster than one 16-row step?  Forward + backward (+ weight gradients), no optimizer; each step replayed from its own
hipGraph; the second replay is held back by a spin kernel so that its encoder forward runs beside the first one's decoder phase."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from gst_visdial_amd import engine as E
from gst_visdial_amd.graph import GraphedStep
dev = torch.device("cuda", 0)
T, R, U, F = 256, 37, 25, 2048

def make(B, seed, own_streams=False):
    model, params = bench.build_model(dev, "bf16", seed=1234)
    model.train()
    V = model.decoder.config.vocab_size
    batch = bench.synthetic_rows(B, T, R, U, F, V, seed, dev)
    if own_streams:
        E._DEVICE_STREAMS.pop(0, None)                     # a fresh (vision, aux) pair for this engine
    def step():
        loss, _ = model(**batch)
        loss.backward()
        for p in model.engine.flat.live:
            p.grad = None
        return loss
    return model, step

def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / n

m16, s16 = make(16, 1)
for _ in range(3): s16()
g16 = GraphedStep(s16, warmup=0)
print("one 16-row step (fwd + bwd + wgrad, no optimizer): %.2f ms" % timeit(g16))
mA, sA = make(8, 2, own_streams=True)
for _ in range(3): sA()
gA = GraphedStep(sA, warmup=0)
mB, sB = make(8, 3, own_streams=True)
for _ in range(3): sB()
gB = GraphedStep(sB, warmup=0)
print("one 8-row step: %.2f ms" % timeit(gA))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(2_000_000); b.record(); torch.cuda.synchronize()
per_ms = 2_000_000 / a.elapsed_time(b)
for delay_ms in (0.0, 1.0, 1.8, 2.5, 3.5):
    def both():
        e = torch.cuda.Event(); e.record()
        s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1):
            gA()
        with torch.cuda.stream(s2):
            if delay_ms: torch.cuda._sleep(int(per_ms * delay_ms))
            gB()
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1); e2.record(s2)
        torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)
    print("two 8-row steps on two stream sets, second delayed %.1f ms: %.2f ms for both" % (delay_ms, timeit(both)))
def both_in_one():
    cur = torch.cuda.current_stream()
    e = torch.cuda.Event(); e.record(cur)
    s1.wait_event(e); s2.wait_event(e)
    with torch.cuda.stream(s1):
        sA()
    with torch.cuda.stream(s2):
        sB()
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    cur.wait_event(e1); cur.wait_event(e2)
for _ in range(2): both_in_one()
torch.cuda.synchronize()
gAB = GraphedStep(both_in_one, warmup=0)
print("two 8-row steps captured into ONE graph on two stream sets: %.2f ms for both" % timeit(gAB))
def serial():
    gA(); gB()
print("two 8-row steps back to back on one stream: %.2f ms" % timeit(serial))
