"""Data-parallel train step with the REAL engine on two ranks (both on cuda:0, collectives over gloo, which accepts
device tensors): the slice-wise all-reduce + 1/N-scaled AdamW of the backward pipeline must leave every rank with the same
parameters as one process that averages the two ranks' gradients by hand.  (RCCL itself is covered with a 1-rank group in
test_model_gpu.py; a node with several GPUs is the driver's.)"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = {0: [0, 1], 1: [2, 1]}          # golden rows each rank trains on
STEPS = 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(sc, g, rows, dev):
    kw = sc.golden_batch(g, dev)
    idx = torch.tensor(rows, device=dev)
    return {k: (v[idx].clone() if torch.is_tensor(v) else v) for k, v in kw.items()}


def _worker(rank, world, port, q, train=False, shard=False, prec="fp32"):
    try:
        _worker_body(rank, world, port, q, train, shard, prec)
    except BaseException as ex:          # noqa: BLE001 -- report instead of dying silently
        import traceback
        q.put((rank, "ERROR: " + "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-1500:]))


def _worker_body(rank, world, port, q, train, shard=False, prec="fp32"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gst_visdial_amd import selfcheck as sc
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.pipeline import BackwardPipeline
    dev = "cuda:0"
    model, params, cfg = sc.build_tiny_model(prec, dev, seed=4)
    model.train(train)
    g = sc.load_npz("tiny_train.npz")
    kw = _batch(sc, g, ROWS[rank], dev)
    opt = FusedAdamW(model, lr=2e-3)
    pipe = BackwardPipeline(model.engine, optimizer=opt, chunk_elems=60000, shard_update=shard)
    losses = []
    for _ in range(STEPS):
        loss, _ = model(**kw)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    torch.cuda.synchronize()
    stale = None
    if shard:
        from gst_visdial_amd._lib import GstvdError
        try:
            model.state_dict()
            stale = "no error"
        except GstvdError as e:
            stale = "sync_master" in str(e)
        pipe.sync_master()                      # fp32 masters / moments of the other rank's shards: gathered on demand
        model.state_dict(); opt.state_dict(); model.encoder.state_dict(); model.decoder.state_dict()
        # what a hipGraph REPLAY of the sharded step does: the device step counter advances, no host code of pipeline.py runs
        # (the guard was a host flag and stayed clear from here on).  Every state_dict() must refuse again.
        opt.step_dev.add_(1.0)
        refused = []
        for fn in (model.state_dict, opt.state_dict, model.encoder.state_dict, model.decoder.state_dict, opt.export_reference_state):
            try:
                fn()
                refused.append(False)
            except GstvdError as e:
                refused.append("sync_master" in str(e))
        # ... and so must an in-place parameter edit, which would re-cast this rank's stale masters over the gathered shadows
        if model.engine.flat.S is not None:
            with torch.no_grad():
                model.vlfusion.fc_v.weight.mul_(1.0)
            try:
                model.engine.flat.refresh_shadow()
                refused.append(False)
            except GstvdError as e:
                refused.append("sync_master" in str(e))
        opt.step_dev.add_(-1.0)
        stale = stale and all(refused)
        pipe.sync_master()
        model.state_dict(); opt.state_dict()
        if model.engine.flat.S is not None:
            model.engine.flat.refresh_shadow()  # masters are current: the re-cast is legal again
        torch.cuda.synchronize()
    mask = None
    if train:           # the keep mask this rank's LAST step drew at one site (ranks must not share a mask stream)
        mask = sc.dropout_keep_masks(model.engine)["t0.ln1"].numpy().copy()
    # numpy, not torch: a tensor would travel as a shared-memory handle that dies with this process
    q.put((rank, losses, model.engine.flat.P.detach().cpu().numpy(), len(pipe.slices), opt.grad_scale, mask,
           int(model.engine.rng.state[0].item()), opt.m.detach().cpu().numpy() if shard else None, stale,
           [(p["S"], p["rest"]) for p in pipe._plans.values()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("train", [False, True], ids=["eval", "train_dropout_on"])
def test_two_rank_pipeline_equals_hand_averaged_gradients(train):
    """train=True: dropout on.  Every rank draws its masks from its own stream (ops.rank_seed: the per-device generators of
    the reference's DataParallel replicas, train_gen.py:295) -- the two ranks' masks differ -- and the result still equals one
    process that runs rank 0's and rank 1's step with those ranks' (seed, offset) states and averages the gradients by hand."""
    sys.path.insert(0, ROOT)
    from gst_visdial_amd import ops, selfcheck as sc
    from gst_visdial_amd.optim import FusedAdamW
    dev = "cuda:0"
    # ---- reference: one process, gradients of the two batches averaged by hand, plain optimizer step
    model, params, cfg = sc.build_tiny_model("fp32", dev, seed=4)
    model.train(train)
    model.engine.prepare(torch.device(dev))

    def as_rank(r, k):      # the dropout state rank r holds in front of its step k (no process group here: set it by hand)
        model.engine.rng.state.copy_(torch.tensor([ops.rank_seed(4, r), k], dtype=torch.int64))
    g = sc.load_npz("tiny_train.npz")
    b0, b1 = _batch(sc, g, ROWS[0], dev), _batch(sc, g, ROWS[1], dev)
    opt = FusedAdamW(model, lr=2e-3)
    ref_losses = []
    for k in range(STEPS):
        as_rank(0, k)
        l0, _ = model(**b0)
        l0.backward()
        g0 = model.engine.flat.G.clone()
        opt.zero_grad()
        as_rank(1, k)
        l1, _ = model(**b1)
        l1.backward()
        model.engine.flat.G.add_(g0).mul_(0.5)
        opt.step()
        opt.zero_grad()
        ref_losses.append((l0.item(), l1.item()))
    torch.cuda.synchronize()
    ref_p = model.engine.flat.P.detach().cpu()
    # ---- two ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, train)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for r in res:
        assert not (isinstance(r[1], str) and r[1].startswith("ERROR")), r[1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, l0, p0, n0, s0, m0, seed0), (r1, l1, p1, n1, s1, m1, seed1) = [r[:7] for r in res]
    assert seed0 == ops.rank_seed(4, 0) == 4 and seed1 == ops.rank_seed(4, 1) != seed0
    if train:
        assert m0.shape == m1.shape and (m0 != m1).mean() > 0.2          # independent Bernoulli(0.7) masks differ in ~42 %
    p0, p1 = torch.from_numpy(p0), torch.from_numpy(p1)
    assert n0 == n1 and n0 >= 3 and s0 == s1 == 0.5                  # several slices, 1/N folded into AdamW
    assert torch.equal(p0, p1)                                       # ranks stay bit-identical
    for i in range(STEPS):
        assert abs(l0[i] - ref_losses[i][0]) < 1e-5 and abs(l1[i] - ref_losses[i][1]) < 1e-5
    assert (p0 - ref_p).abs().max().item() < 2e-6


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_two_rank_sharded_update_equals_hand_averaged_gradients(prec):
    """BackwardPipeline(shard_update=True) with the real engine on two ranks (gloo over device tensors): reduce-scatter -> AdamW on
    the rank's half of every slice -> all-gather of the bf16 shadow weights + the fp32-read parameters (bf16 mode) or of the fp32
    weights (fp32 mode).  The ranks stay bit-identical in everything the forward reads, the losses follow the hand-averaged
    single-process run, and after sync_master() so do master weights and moments; state_dict() before it refuses loudly."""
    sys.path.insert(0, ROOT)
    from gst_visdial_amd import ops, selfcheck as sc
    from gst_visdial_amd.optim import FusedAdamW
    dev = "cuda:0"
    model, params, cfg = sc.build_tiny_model(prec, dev, seed=4)
    model.eval()
    model.engine.prepare(torch.device(dev))
    g = sc.load_npz("tiny_train.npz")
    b0, b1 = _batch(sc, g, ROWS[0], dev), _batch(sc, g, ROWS[1], dev)
    opt = FusedAdamW(model, lr=2e-3)
    ref_losses = []
    for k in range(STEPS):
        l0, _ = model(**b0)
        l0.backward()
        g0 = model.engine.flat.G.clone()
        opt.zero_grad()
        l1, _ = model(**b1)
        l1.backward()
        model.engine.flat.G.add_(g0).mul_(0.5)
        opt.step()
        opt.zero_grad()
        ref_losses.append((l0.item(), l1.item()))
    torch.cuda.synchronize()
    ref_p, ref_m = model.engine.flat.P.detach().cpu(), opt.m.detach().cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, False, True, prec)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for r in res:
        assert not (isinstance(r[1], str) and r[1].startswith("ERROR")), r[1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = res
    p0, p1 = torch.from_numpy(a[2]), torch.from_numpy(b[2])
    m0, m1 = torch.from_numpy(a[7]), torch.from_numpy(b[7])
    assert a[8] is True and b[8] is True                           # state_dict() before sync_master(): refused, message names the way out
    assert any(S > 0 for S, rest in a[9]) and any(rest > 0 for S, rest in a[9])      # both the sharded bulk and a replicated rest were exercised
    assert torch.equal(p0, p1) and torch.equal(m0, m1)             # after sync_master() the ranks hold the same complete state
    tol_l, tol_p = (1e-5, 2e-6) if prec == "fp32" else (2e-3, 5e-5)
    for i in range(STEPS):
        assert abs(a[1][i] - ref_losses[i][0]) < tol_l and abs(b[1][i] - ref_losses[i][1]) < tol_l
    assert (p0 - ref_p).abs().max().item() < tol_p and (m0 - ref_m).abs().max().item() < tol_p
