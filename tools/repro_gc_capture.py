"""Root-cause probe for the round-1 SIGABRT in the hipGraph-captured train step (GPUTEST_r01: rc 134 at graph.py:40).

Hypothesis: torch >= 2.9 no longer runs gc.collect() in torch.cuda.graph.__enter__, so an automatic cyclic-GC pass can
fire in the middle of a stream capture.  If it finds a dead Engine <-> model cycle from an earlier test that still owns
a decode session (19 CUDAGraph objects with a private memory pool), ~CUDAGraph releases the pool (hipFree &c.) while a
capture in "global" error mode is open -> the call is refused, the destructor throws, std::terminate -> SIGABRT.

    python tools/repro_gc_capture.py raw      # garbage + gc.collect() forced inside a raw torch.cuda.graph capture
    python tools/repro_gc_capture.py guarded  # same garbage, capture through gst_visdial_amd.graph.GraphedStep

Run each mode in its own process and read the exit code (134 = reproduced)."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gst_visdial_amd import selfcheck as s          # noqa: E402
from gst_visdial_amd.optim import FusedAdamW        # noqa: E402

DEV = "cuda:0"


def make_garbage():
    """A model whose engine owns a captured decode session, dropped without being collected."""
    g = s.load_npz("tiny_train.npz")
    model, params, cfg = s.build_tiny_model("fp32", DEV, mode="vd_gen_val")
    model.eval()
    kw = s.golden_batch(g, DEV)
    kw["dec_input_ids"] = torch.full((kw["enc_input_ids"].shape[0], 1), 101, dtype=torch.long, device=DEV)
    kw["dec_labels"] = None
    model(temperature=0.7, top_k=7, **kw)
    assert len(model.engine._decode_sessions) == 1
    # what round 1's Engine <-> model reference cycle amounted to (the engine now holds its model weakly): an unreachable
    # cycle that owns the model, its engine and the engine's captured decode session
    holder = {"model": model}
    holder["self"] = holder
    return True


def main(mode):
    gc.disable()                       # keep the garbage until we decide
    cyc = make_garbage()
    print("garbage made (engine<->model strong cycle: %s)" % cyc, flush=True)
    g = s.load_npz("tiny_train.npz")
    model, params, cfg = s.build_tiny_model("fp32", DEV, seed=7)
    model.train()
    kw = s.golden_batch(g, DEV)
    opt = FusedAdamW(model, lr=1e-3)
    state = dict(collect=False)

    def one():
        loss, _ = model(**kw)
        if state["collect"]:
            n = gc.collect()           # what an automatic GC pass would do at this point
            print("gc.collect() inside the capture freed %d objects" % n, flush=True)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    one(); one()
    torch.cuda.synchronize()
    state["collect"] = True
    if mode == "raw":
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="global"):
            one()
    else:
        from gst_visdial_amd.graph import GraphedStep
        step = GraphedStep(one, warmup=0)
        graph = step.graph
    state["collect"] = False
    graph.replay()
    torch.cuda.synchronize()
    print("capture + replay ok (%s)" % mode, flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "raw")
