"""Sharded optimizer update of the data-parallel path (gst_visdial_amd/pipeline.py, BackwardPipeline(shard_update=True)) on CPU
with gloo, world sizes 2 and 8: per slice reduce-scatter -> AdamW on the rank's 1/N shard -> all-gather of the bf16 shadow weights
(+ the fp32-read parameters).  The reference's optimizer runs once, on GPU 0 (train_gen.py:326-329); the all-reduce path of this
package runs it in full on every rank.  Checked here, without a GPU: every rank ends with the same forward-visible state as the
all-reduce path (bit for bit where the collectives sum in the same order, to fp32 rounding otherwise), the master weights and
moments are complete after sync_master(), and the partition tiles every slice exactly once.

The optimizer below is a HOST stand-in with FusedAdamW's interface (the product's AdamW is a HIP kernel and has no CPU path): it is
test infrastructure, the pipeline under test is the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Flat:
    def __init__(self, n, shadow=True):
        g = torch.Generator().manual_seed(5)
        self.P = torch.randn(n, generator=g)
        self.G = torch.zeros(n)
        self.S = self.P.to(torch.bfloat16) if shadow else None
        self.n_live = n
        self.shadow_version = None

    def version(self):
        return 0

    def fp32_read_ranges(self):
        # "biases / LayerNorm / embedding table": a large range at the start, small ones sprinkled through the buffer
        n = self.n_live
        r = [(0, n // 7)]
        for k in range(1, 40):
            a = n // 7 + k * (n // 50)
            if a + 97 < n:
                r.append((a, a + 97))
        return r


class _Engine:
    def __init__(self, n, shadow=True):
        self.flat = _Flat(n, shadow)
        self.pipe = None


class _HostAdamW:
    """pytorch_transformers AdamW semantics (optim.py) on host tensors, FusedAdamW's apply_range interface."""

    def __init__(self, engine, lr=1e-2, wd=0.01, betas=(0.9, 0.999), eps=1e-6):
        self.engine, self.lr, self.wd, self.betas, self.eps = engine, lr, wd, betas, eps
        self.m, self.v = torch.zeros_like(engine.flat.P), torch.zeros_like(engine.flat.P)
        self.grad_scale, self.t = 1.0, 0
        self._applied_in_backward = False
        self.calls = []

    def begin_step(self):
        self.t += 1

    def _sync_step(self):          # FusedAdamW: the DEVICE step counter (it advances inside a replayed hipGraph)
        return self.t

    def apply_range(self, lo, hi, grad_bf16=None, fused=(), grad_origin=None):
        f = self.engine.flat
        self.calls.append((lo, hi))
        if grad_bf16 is not None:
            o = lo if grad_origin is None else grad_origin
            g = grad_bf16[lo - o:hi - o].float()
        else:
            g = f.G[lo:hi]
        g = g * self.grad_scale
        b1, b2 = self.betas
        m, v, p = self.m[lo:hi], self.v[lo:hi], f.P[lo:hi]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc = (1 - b2 ** self.t) ** 0.5 / (1 - b1 ** self.t)
        p.addcdiv_(m, v.sqrt().add_(self.eps), value=-self.lr * bc)
        p.mul_(1 - self.lr * self.wd)
        if f.S is not None:
            f.S[lo:hi].copy_(p)


def _grads(n, rank, step):
    return torch.randn(n, generator=torch.Generator().manual_seed(1000 * step + rank)) * 0.1


def _train(world, rank, n, marks, chunk, compress, shard, shadow, steps):
    from gst_visdial_amd.pipeline import BackwardPipeline
    eng = _Engine(n, shadow)
    opt = _HostAdamW(eng)
    pipe = BackwardPipeline(eng, optimizer=opt, chunk_elems=chunk, compress=compress, shard_update=shard)
    assert pipe.collective and pipe.world == world and opt.grad_scale == 1.0 / world
    for k in range(steps):
        eng.flat.G.copy_(_grads(n, rank, k))
        pipe.begin()
        for off in marks:
            if pipe.ready(off):
                pipe.run_slice(off, pipe.hi)
        if pipe.hi > 0:
            pipe.run_slice(0, pipe.hi)
        pipe.end()
    return eng, opt, pipe


def _worker(rank, world, port, n, marks, chunk, compress, shadow, steps, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ref_eng, ref_opt, ref_pipe = _train(world, rank, n, marks, chunk, compress, False, shadow, steps)
        eng, opt, pipe = _train(world, rank, n, marks, chunk, compress, True, shadow, steps)
        f, rf = eng.flat, ref_eng.flat
        # what the forward reads: the bf16 shadow everywhere (or fp32 P without one) + fp32 P on the fp32-read ranges
        vis = f.S.float() if shadow else f.P.clone()
        ref_vis = rf.S.float() if shadow else rf.P.clone()
        fp32_idx = torch.cat([torch.arange(a, b) for a, b in f.fp32_read_ranges()])
        vis_p, ref_vis_p = f.P[fp32_idx].clone(), rf.P[fp32_idx].clone()
        # the partition: every element of every slice updated exactly once per step by SOME rank -> count the calls of this rank
        cover = torch.zeros(n, dtype=torch.int32)
        for lo, hi in opt.calls:
            cover[lo:hi] += 1
        stale_before = (f.P != rf.P).float().mean().item() if world > 1 else 0.0
        flags = [pipe.master_stale]
        pipe.sync_master()
        flags.append(pipe.master_stale)
        # a hipGraph REPLAY of a sharded step: the captured kernels advance the optimizer's device step counter, none of
        # pipeline.py's host code runs (a host flag set in _run_sharded stayed False from here on)
        opt.t += 1
        flags.append(pipe.master_stale)
        try:
            pipe.check_master_current("state_dict()")
            flags.append("no error")
        except Exception as e:      # noqa: BLE001
            flags.append("sync_master" in str(e))
        opt.t -= 1
        out = dict(rank=rank, slices=list(pipe.slices), vis=vis.numpy(), ref_vis=ref_vis.numpy(), vis_p=vis_p.numpy(), ref_vis_p=ref_vis_p.numpy(),
                   P=f.P.numpy().copy(), m=opt.m.numpy().copy(), v=opt.v.numpy().copy(), refP=rf.P.numpy().copy(), refm=ref_opt.m.numpy().copy(),
                   cover=cover.numpy(), stale_before=stale_before, stale_flags=flags, plans={k: (p["S"], p["bulk"], p["rest"], p["a"], p["b"]) for k, p in pipe._plans.items()})
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as ex:   # noqa: BLE001
        import traceback
        q.put(dict(rank=rank, error="".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-2000:]))


def _spawn(world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert "error" not in r, r["error"]
    for p in procs:
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,compress,shadow,slice_min", [(2, None, True, None), (2, "bf16", True, 2048), (8, "bf16", True, None), (2, None, False, None)],
                         ids=["w2_fp32_payload", "w2_bf16_payload_slice_copies", "w8_bf16_payload", "w2_fp32_precision_no_shadow"])
def test_sharded_update_matches_allreduce_path(world, compress, shadow, slice_min, monkeypatch):
    import numpy as np
    if slice_min is not None:        # (the workers are spawned: they read the threshold from the environment) the large fp32-read
        monkeypatch.setenv("GSTVD_PACK_SLICE_MIN", str(slice_min))      # range of the stub then travels as slice copies, the small ones by index
    n, steps = 100000, 3
    graded = [30000, 25000, 12000, 6000]
    marks = [90000, 69000, 52000, 44000, 30000, 21000, 12500, 6000, 500, 0]
    res = _spawn(world, (n, marks, graded, compress, shadow, steps))
    r0 = res[0]
    # --- the partition
    S_align = 1024
    tot = np.zeros(n, dtype=np.int64)
    for r in res:
        assert r["slices"] == r0["slices"] and r["slices"][0][1] == n and r["slices"][-1][0] == 0
        tot += r["cover"]
        for (lo, hi), (S, bulk, rest, a, b) in r["plans"].items():
            assert S % S_align == 0 and bulk == world * S and 0 <= rest < world * S_align and bulk + rest == hi - lo
            assert a == lo + r["rank"] * S and b == a + S
    # every element: `steps` updates by its owner (bulk) or `steps` updates by EVERY rank (rest of a slice)
    in_rest = np.zeros(n, dtype=bool)
    for (lo, hi), (S, bulk, rest, a, b) in r0["plans"].items():
        in_rest[lo + bulk:hi] = True
    assert (tot[~in_rest] == steps).all() and (tot[in_rest] == steps * world).all()
    assert any(p[0] > 0 for p in r0["plans"].values()) and in_rest.any()
    # --- ranks agree bit for bit on everything the forward reads, and (after sync_master) on master weights and moments
    for r in res:      # stale before sync_master(), current after it, stale again after a (simulated) graph replay, and the guard raises
        assert r["stale_flags"] == [True, False, True, True], r["stale_flags"]
    for r in res[1:]:
        assert (r["vis"] == r0["vis"]).all() and (r["vis_p"] == r0["vis_p"]).all()
        assert (r["P"] == r0["P"]).all() and (r["m"] == r0["m"]).all() and (r["v"] == r0["v"]).all()
    if shadow:
        assert r0["stale_before"] > 0.3        # (the master weights of other ranks' shards really were stale before the gather)
    else:
        assert r0["stale_before"] <= 1e-3      # fp32 precision mode: the fp32 weights themselves are what the all-gather moves
    # --- and with the all-reduce path: same sums, possibly in another order
    tol = 2e-2 if compress == "bf16" else 1e-5
    for r in res:
        assert np.abs(r["P"] - r["refP"]).max() <= tol * max(1.0, np.abs(r["refP"]).max())
        assert np.abs(r["m"] - r["refm"]).max() <= tol
        assert np.abs(r["vis"] - r["ref_vis"]).max() <= max(tol, 2e-2 if shadow else tol) * max(1.0, np.abs(r["ref_vis"]).max())
        assert np.abs(r["vis_p"] - r["ref_vis_p"]).max() <= tol * max(1.0, np.abs(r["ref_vis_p"]).max())
    if compress is None and world == 2:
        # two ranks, fp32 payload: a + b in either order is the same float -> bit-identical to the all-reduce path
        assert (r0["P"] == r0["refP"]).all() and (r0["m"] == r0["refm"]).all()


def test_skip_update_once_keeps_local_gradients_and_applies_the_sum_next_time():
    """train_gen.py:326-329: iteration 0 neither steps nor zeroes the gradients.  pipe.skip_update_once(): that backward runs no
    collective and no update; the next backward's (accumulated) gradients are reduced and applied once."""
    from gst_visdial_amd.pipeline import BackwardPipeline
    n = 5000
    eng = _Engine(n)
    opt = _HostAdamW(eng)
    pipe = BackwardPipeline(eng, optimizer=opt, chunk_elems=2000)
    p0 = eng.flat.P.clone()
    g0, g1 = _grads(n, 0, 0), _grads(n, 0, 1)
    pipe.skip_update_once()
    eng.flat.G.copy_(g0)
    pipe.begin()
    assert pipe.fuse_handle() is None
    for off in (3000, 1000, 0):
        if pipe.ready(off):
            pipe.run_slice(off, pipe.hi)
    pipe.end()
    assert torch.equal(eng.flat.P, p0) and opt.t == 0 and not opt._applied_in_backward and opt.calls == []
    assert torch.equal(eng.flat.G, g0)                       # the local gradients are still there
    eng.flat.G.add_(g1)                                      # what the engine's accumulate mode does at iteration 1
    pipe.begin()
    for off in (3000, 1000, 0):
        if pipe.ready(off):
            pipe.run_slice(off, pipe.hi)
    pipe.end()
    assert opt.t == 1 and opt._applied_in_backward and not torch.equal(eng.flat.P, p0)
    ref = _Engine(n)
    ropt = _HostAdamW(ref)
    ref.flat.G.copy_(g0 + g1)
    ropt.begin_step()
    ropt.apply_range(0, n)
    assert torch.allclose(eng.flat.P, ref.flat.P, atol=0, rtol=0)
