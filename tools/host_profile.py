#!/usr/bin/env python3
"""Where the host time of an EAGERLY issued train step goes (658 launching calls, 18.7 ms of host time per
step): cProfile of 5 eager steps of the bench's step function at the full configuration, sorted by own time and by cumulative time,
plus the raw cost of the pieces a call is made of (ctypes call of a trivial entry point, data_ptr(), current_stream().cuda_stream).
usage: python3 tools/host_profile.py"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from gst_visdial_amd.optim import FusedAdamW
from gst_visdial_amd.pipeline import BackwardPipeline
from gst_visdial_amd import _lib as L

dev = torch.device("cuda:0")
model, params = bench.build_model(dev, "bf16", seed=1234)
model.train()
V = model.decoder.config.vocab_size
batch = bench.synthetic_rows(16, 256, 37, 25, 2048, V, 1234, dev)
opt = FusedAdamW(model, lr=2e-5, warmup_steps=1500, t_total=100000)
pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=1 << 40)


def step():
    loss, _ = model(**batch)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("eager issue: %.2f ms of host time per step (5 steps, no sync inside)" % ((t1 - t0) * 1e3 / 5))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:44]))
lib = L.load()
n = 20000
t0 = time.perf_counter()
for _ in range(n):
    lib.gstvd_abi_version()
print("ctypes call of a trivial entry point: %.2f us" % ((time.perf_counter() - t0) * 1e6 / n))
x = torch.zeros(4, device=dev)
t0 = time.perf_counter()
for _ in range(n):
    x.data_ptr()
print("tensor.data_ptr(): %.2f us" % ((time.perf_counter() - t0) * 1e6 / n))
t0 = time.perf_counter()
for _ in range(n):
    torch.cuda.current_stream().cuda_stream
print("torch.cuda.current_stream().cuda_stream: %.2f us" % ((time.perf_counter() - t0) * 1e6 / n))
t0 = time.perf_counter()
for _ in range(2000):
    lib.gstvd_rng_advance(model.engine.rng.state.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("one tiny kernel launch through ctypes (gstvd_rng_advance): %.2f us" % ((time.perf_counter() - t0) * 1e6 / 2000))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    x.zero_()
print("torch op (x.zero_()): %.2f us" % ((time.perf_counter() - t0) * 1e6 / 2000))
torch.cuda.synchronize()
