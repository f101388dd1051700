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


# ---- N > 1 with a bf16 payload: the weight-gradient launch writes the payload itself -------------------------
def test_cast_ranges_matches_the_plain_cast_on_its_ranges_and_leaves_the_rest_alone():
    from gst_visdial_amd import ops as o
    n = 300000
    g = torch.Generator().manual_seed(1)
    src = torch.randn(n, generator=g).to(DEV)
    ranges = [(0, 4), (64, 1000), (2048, 1024), (8192, 5), (65536, 70000), (n - 12, 12)]
    dst = torch.full((n,), 7.0, device=DEV, dtype=torch.bfloat16)
    plan = o.CastRanges(ranges, torch.device(DEV))
    plan.run(src, dst)
    want = torch.full((n,), 7.0, device=DEV, dtype=torch.bfloat16)
    ref = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    o.cast(src, ref)
    for a, m in ranges:
        want[a:a + m] = ref[a:a + m]
    assert torch.equal(dst, want) and plan.elems == sum(m for _, m in ranges)
    o.CastRanges([], torch.device(DEV)).run(src, dst)             # nothing to do is not an error
    with pytest.raises(Exception):
        o.CastRanges([(2, 8)], torch.device(DEV))


@pytest.mark.parametrize("shard", [False, True], ids=["allreduce", "sharded_update"])
def test_direct_bf16_payload_is_bit_identical_to_store_then_cast(shard):
    """bf16 mode, 1-rank RCCL group: with direct_bf16 the grouped weight-gradient launch writes the all-reduce payload of every GEMM
    weight in bf16 and only the rest of a slice is cast; the rounding is the cast's, so parameters, moments and shadows are BIT-identical
    to the store-fp32-then-cast path -- eagerly and under graph replay.  `.grad` of a weight whose fp32 gradient was never stored is None."""
    import torch.distributed as dist
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    from gst_visdial_amd.graph import GraphedStep
    s = sc()
    g = load_npz("tiny_train.npz")
    created = _group()
    try:
        def run(direct, graphed):
            model, params, cfg = s.build_tiny_model("bf16", DEV, seed=3, cfg_file="tiny_cfg_dropout.json")
            model.train()
            kw = s.golden_batch(g, DEV)
            opt = FusedAdamW(model, lr=1e-3)
            pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=100000, compress="bf16", force_collective=True, shard_update=shard,
                                    direct_bf16=direct)

            def step():
                loss, _ = model(**kw)
                loss.backward()
                none = sum(1 for p in model.engine.flat.live if p.grad is None)
                opt.step()
                opt.zero_grad()
                return loss, none

            for _ in range(2):
                loss, none = step()
            fn = (lambda f=GraphedStep(lambda: step()[0], warmup=0): f()) if graphed else (lambda: step()[0])
            for _ in range(3):
                loss = fn()
            torch.cuda.synchronize()
            eng = model.engine
            return loss.item(), eng.flat.P.clone(), eng.flat.S.clone(), opt.m.clone(), opt.v.clone(), none, pipe

        ref = run(False, False)
        for graphed in (False, True):
            got = run(True, graphed)
            assert got[0] == ref[0]
            for a, b in zip(ref[1:5], got[1:5]):
                assert torch.equal(a, b)
        assert ref[5] == 0 and got[5] > 20                       # the GEMM weights' `.grad` is None in direct mode (never stored in fp32)
        pipe = got[6]
        assert pipe.Gb is not None and any(len(k[2]) > 0 for k in pipe._cast_plans)
        cast_elems = sum(pl[0].elems for pl in pipe._cast_plans.values())
        assert cast_elems < 0.6 * pipe.Gb.numel()               # (tiny model: embeddings dominate; at full size the cast covers ~7 %)
    finally:
        if created:
            dist.destroy_process_group()
