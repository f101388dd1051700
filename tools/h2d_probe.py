"""Where does the PCIe-inclusive step time go?  Replay of the captured train step with its inputs refreshed from the host in
these ways: not at all, pageable .to(device), PinnedStager one batch ahead, PinnedStager without prefetch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from gst_visdial_amd.optim import FusedAdamW
from gst_visdial_amd.pipeline import BackwardPipeline
from gst_visdial_amd.graph import GraphedStep
from gst_visdial_amd.step import PinnedStager
dev = torch.device("cuda", 0)
B, T, R, U, F = 16, 256, 37, 25, 2048
model, params = bench.build_model(dev, "bf16", seed=1234)
V = model.decoder.config.vocab_size
model.train()
batch = bench.synthetic_rows(B, T, R, U, F, V, 1234, dev)
opt = FusedAdamW(model, lr=2e-5, warmup_steps=1500, t_total=100000)
BackwardPipeline(model.engine, optimizer=opt, chunk_elems=192 << 20)
def device_step():
    loss, _ = model(**batch); loss.backward(); opt.step(); opt.zero_grad(); return loss
for _ in range(3): device_step()
replay = GraphedStep(device_step, warmup=0)
for _ in range(3): replay()
torch.cuda.synchronize()
def timeit(name, fn, n=15):
    fn(2); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
    print("%-40s %.2f ms/step" % (name, (time.perf_counter() - t) * 1e3 / n), flush=True)
timeit("replay only", lambda n: [replay() for _ in range(n)])
extra = {k: v.clone() for k, v in batch.items()}
def f(n):
    for _ in range(n):
        for k, v in extra.items(): batch[k].copy_(v)
        replay()
timeit("9 D2D copies + replay", f)
host = [{k: v.cpu() for k, v in bench.synthetic_rows(B, T, R, U, F, V, 4000 + i, "cpu").items()} for i in range(3)]
def g(n):
    for i in range(n):
        rows = {k: v.to(dev) for k, v in host[i % 3].items()}
        for k, v in rows.items(): batch[k].copy_(v)
        replay()
timeit("pageable .to(dev) + copies + replay", g)
mode = next((a[7:] for a in sys.argv if a.startswith("--mode=")), "pinned")
st = PinnedStager(dev, depth=2, mode=mode)
print("stager mode =", mode, "| torch threads", torch.get_num_threads())
def h(n):
    pending = st.put(host[0])
    for i in range(n):
        rows = st.get(pending)
        for k, v in rows.items(): batch[k].copy_(v)
        replay()
        pending = st.put(host[(i + 1) % 3])        # host fill of the next slot while the replay runs
    st.get(pending)
timeit("stager prefetch + copies + replay", h)
def h2(n):
    for i in range(n):
        t0 = time.perf_counter(); hd = st.put(host[i % 3]); t1 = time.perf_counter(); rows = st.get(hd); t2 = time.perf_counter()
        for k, v in rows.items(): batch[k].copy_(v)
        t3 = time.perf_counter(); replay(); t4 = time.perf_counter()
        if i == n - 1: print("   host ms: put %.2f get %.2f copies %.2f replay-launch %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
timeit("stager put+get (no prefetch)", h2)
