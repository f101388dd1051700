"""Round 6 on the GPU: the SEGMENTED replay of the N > 1 step (graph.SegmentedStep: one hipGraph per gradient slice, the slices'
collectives and updates issued eagerly between the replays) on a 1-rank RCCL group -- the fall-back for a refused whole-step
capture.  Parity = the state of the whole-step graph, which tests/test_model_gpu.py ties to eager issue."""
import os

import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sc():
    from gst_visdial_amd import selfcheck
    return selfcheck


def _group():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    return created


@pytest.mark.parametrize("train,compress,shard", [(False, "bf16", False), (True, "bf16", False), (True, None, False), (True, "bf16", True)],
                         ids=["eval_bf16_payload", "train_bf16_payload", "train_fp32_payload", "train_sharded_update"])
def test_segmented_replay_equals_the_whole_step_graph(train, compress, shard):
    import torch.distributed as dist
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    from gst_visdial_amd.graph import GraphedStep, SegmentedStep
    s = sc()
    g = load_npz("tiny_train.npz")
    created = _group()
    try:
        def run(kind, steps=4):
            model, params, cfg = s.build_tiny_model("fp32", DEV, seed=3, cfg_file="tiny_cfg_dropout.json" if train else "tiny_cfg.json")
            model.train(train)
            kw = s.golden_batch(g, DEV)
            opt = FusedAdamW(model, lr=1e-3)
            pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=100000, compress=compress, force_collective=True,
                                    shard_update=shard)

            def step():
                loss, _ = model(**kw)
                loss.backward()
                opt.step()
                opt.zero_grad()
                return loss

            for _ in range(2):
                step()
            fn = GraphedStep(step, warmup=0) if kind == "graph" else SegmentedStep(step, pipe, warmup=0)
            losses = []
            for _ in range(steps):
                losses.append(fn().item())
            torch.cuda.synchronize()
            eng = model.engine
            return (losses, eng.flat.P.clone(), opt.m.clone(), opt.v.clone(), opt._sync_step(), int(eng.rng.state[1].item()), fn, pipe)

        lg, pg, mg, vg, tg, og, _, _ = run("graph")
        ls, ps, ms, vs, ts, os_, seg, pipe = run("segmented")
        nsl = len(pipe.slices)
        assert nsl >= 3                                                     # several cuts, not one
        kinds = [k for k, _ in seg.items]
        assert seg.n_graphs == nsl + 1 and kinds.count("call") == nsl + 1   # a graph in front of every slice's collective + the tail; + the final join
        assert kinds[0] == "graph" and kinds[-1] == "call" and kinds[-2] == "graph"
        assert tg == ts == 6 and og == os_                                  # optimizer step counter and dropout offset advanced inside the replays
        assert all(abs(a - b) <= 2e-6 * max(1.0, abs(a)) for a, b in zip(ls, lg)), (ls, lg)      # (ulp-level: see the atomics below)
        # same kernels on the same data; the embedding tables' gradients are atomic scatter-adds (their order is not fixed)
        for a, b in ((pg, ps), (mg, ms), (vg, vs)):
            assert (a - b).abs().max().item() < 1e-6
    finally:
        if created:
            dist.destroy_process_group()


def test_segmented_step_refuses_a_pipeline_without_a_collective():
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    from gst_visdial_amd.graph import SegmentedStep
    model, params, cfg = sc().build_tiny_model("fp32", DEV, seed=3)
    opt = FusedAdamW(model, lr=1e-3)
    pipe = BackwardPipeline(model.engine, optimizer=opt)
    with pytest.raises(ValueError):
        SegmentedStep(lambda: None, pipe)
