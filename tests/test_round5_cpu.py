"""Round-5 host logic that needs no GPU: the XCD placement of a grouped launch's tiles, the fusion guard of the weight-gradient
queue, the refusal of a replicated model, and the documented ABI version."""
import json
import os
import re
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _step_table():
    """Weight-gradient problems of one 16-row train step, (M, N, K) = (out features, in features, batch rows), backward order."""
    shapes = [(30528, 768, 400)]
    for _ in range(12):
        shapes += [(768, 3072, 400), (3072, 768, 400), (768, 768, 400), (768, 768, 400), (768, 768, 400), (2304, 768, 400)]
    shapes += [(18432, 768, 4688), (768, 1024, 592), (768, 768, 4096)]
    for _ in range(6):
        shapes += [(768, 3072, 4096), (3072, 768, 4096), (768, 1024, 4096), (3072, 768, 4096)]
        shapes += [(1024, 1024, 592)] * 3 + [(3072, 1024, 592)] + [(1024, 1024, 592)] * 3 + [(3072, 1024, 592)]
        for _ in range(2):
            shapes += [(768, 3072, 4096), (3072, 768, 4096), (768, 768, 4096), (2304, 768, 4096)]
    shapes += [(1024, 2048, 592)]
    return shapes


@pytest.mark.parametrize("mode", [1])
@pytest.mark.parametrize("fused", [True, False])
def test_xcd_block_map_is_a_placement_of_every_tile_exactly_once(mode, fused):
    from gst_visdial_amd.ops import xcd_block_map, N_XCD
    shapes = _step_table()
    T = 256
    per = [((M + T - 1) // T) * ((N + T - 1) // T) for M, N, K in shapes]
    total = sum(per)
    bm = xcd_block_map(shapes, T, fused, mode)
    assert len(bm) % N_XCD == 0 and len(bm) >= total
    live = [t for t in bm if t >= 0]
    assert sorted(live) == list(range(total))                      # a bijection onto the tiles; the rest is padding
    assert all(t == -1 for t in bm if t < 0) and len(bm) - total < 0.05 * total
    # tile id -> (problem, K)
    owner, kk = [], []
    for i, n in enumerate(per):
        owner += [i] * n
        kk += [shapes[i][2]] * n
    kmax = max(kk)
    for x in range(N_XCD):
        q = [t for t in bm[x::N_XCD] if t >= 0]
        # (1) a problem's tiles sit on ONE XCD in runs of consecutive ids (whole problems, or <= 40-tile pieces of the large ones)
        runs, start = [], 0
        for j in range(1, len(q) + 1):
            if j == len(q) or q[j] != q[j - 1] + 1:
                runs.append(q[start:j]); start = j
        for r in runs:
            assert len(set(owner[t] for t in r)) <= 2 or len(r) <= 40
        small = [i for i in set(owner[t] for t in q) if per[i] <= 40]
        for i in small:
            assert sum(1 for t in q if owner[t] == i) == per[i]     # a small problem is never split over XCDs
        # (2) long-K tiles come first and in one block: equal-K tiles that start together stay together
        is_long = [2 * kk[t] > kmax for t in q]
        first_long = is_long.index(True)
        last_long = len(is_long) - 1 - is_long[::-1].index(True)
        assert all(is_long[first_long:last_long + 1])
        assert first_long == 0
    # (3) the queues carry about the same work
    def cost(t):
        return (35.0 if fused else 14.0) + 0.76 * ((kk[t] + 31) // 32)
    loads = [sum(cost(t) for t in bm[x::N_XCD] if t >= 0) for x in range(N_XCD)]
    assert max(loads) / min(loads) < 1.03


def test_small_tables_are_left_to_the_library_order():
    from gst_visdial_amd import ops
    bm = ops.xcd_block_map([(768, 768, 400)] * 3, 256, True, 3)
    assert sorted(t for t in bm if t >= 0) == list(range(27))
    with pytest.raises(ValueError):          # round 5's other candidates (staggered lead-in, 30-tile pieces) are gone
        ops.xcd_block_map([(768, 768, 400)] * 3, 256, True, 2)


def test_queue_refuses_fusion_for_overlapping_gradient_blocks():
    """a weight whose gradient block is also written by ANOTHER queued problem (accumulating or not) is not the only
    contribution of the step -- the launch must not update it in its epilogue."""
    from gst_visdial_amd.ops import GemmGroup
    base = 1 << 20
    def item(c, M, N, acc):
        return (0, 0, c, M, N, 64, M, N, N, acc, 0, 0)
    items = [item(base, 64, 32, 0), item(base + 4 * 64 * 32, 16, 32, 0),            # disjoint neighbours
             item(base + 100000, 64, 32, 0), item(base + 100000, 64, 32, 1),         # same block twice (second accumulates)
             item(base + 300000, 64, 32, 0), item(base + 300000 + 4 * 32 * 10, 8, 32, 0)]   # a block inside another
    assert GemmGroup._overlapping(items) == {2, 3, 4, 5}
    assert GemmGroup._overlapping(items[:2]) == set()


def _tiny_model():
    from gst_visdial_amd.modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    d = tempfile.mkdtemp()
    json.dump(cfg["enc"], open(d + "/e.json", "w"))
    json.dump(cfg["dec"], open(d + "/d.json", "w"))
    params = dict(model_enc_config=d + "/e.json", model_dec_config=d + "/d.json", gpu_ids=[0], model="enc_dec_a", mode="vd_train", batch_size=3)
    enc, dec = VisualDialogEncoder(params), VisualDialogDecoder(params)
    return EncoderDecoderModel(params, enc, dec), enc, dec


def test_replication_over_several_devices_is_refused_loudly():
    """train_gen.py:295 / README.md:89 document nn.DataParallel(model, [0, 1, 2, 3]); replicas would share one engine.  The
    replicate path raises with the way out in the message; a single-id DataParallel never replicates (covered on the GPU)."""
    from gst_visdial_amd._lib import GstvdError
    model, enc, dec = _tiny_model()
    with pytest.raises(GstvdError) as e:
        model._replicate_for_data_parallel()
    assert "torchrun" in str(e.value) and "BackwardPipeline" in str(e.value)
    with pytest.raises(GstvdError):
        torch.nn.parallel.replicate(model, [0, 1]) if torch.cuda.device_count() >= 2 else model._replicate_for_data_parallel()
    dp = torch.nn.DataParallel(model, device_ids=[0]) if torch.cuda.is_available() else None     # construction itself stays legal
    assert dp is None or dp.module is model


def test_fp32_read_ranges_cover_everything_but_pure_gemm_weights():
    from gst_visdial_amd.engine import FlatParams
    model, enc, dec = _tiny_model()
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings
    fp = FlatParams(model, "bf16")
    rs = fp.fp32_read_ranges()
    assert all(a < b for a, b in rs) and all(b <= c for (_, b), (c, _) in zip(rs, rs[1:])) and rs[-1][1] <= fp.n_live

    def inside(off, n):
        return any(a <= off and off + n <= b for a, b in rs)

    def outside(off, n):
        return all(off + n <= a or b <= off for a, b in rs)
    for name, (off, shape) in fp.slots.items():
        n = 1
        for d in shape:
            n *= d
        if name.endswith(".b") or ".ln" in name or name.endswith((".vln.w", ".tln.w")) or name.startswith("emb.") or name == "vemb.loc.w":
            assert inside(off, n), name
        elif name.endswith(".w"):
            assert outside(off, n), name
    # the tied LM head (no train_gen.py:293 aliasing): its weight IS the decoder's word embedding table -> read in fp32 too
    model2, enc2, dec2 = _tiny_model()
    fp2 = FlatParams(model2, "bf16")
    assert fp2.slots["lm.w"] == fp2.slots["demb.word"]
    off, shape = fp2.slots["lm.w"]
    assert any(a <= off and off + shape[0] * shape[1] <= b for a, b in fp2.fp32_read_ranges())


def test_documented_abi_version_is_the_librarys():
    """INTEGRATION.md's ctypes stub asserted ABI 3 while the library was at 4."""
    from gst_visdial_amd import _lib
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"gstvd_abi_version\(\)\s*==\s*(\d+)", txt)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    src = open(os.path.join(ROOT, "gst_visdial_amd", "csrc", "loss.hip")).read()
    m = re.search(r"gstvd_abi_version\(void\)\s*\{\s*return\s+(\d+);", src)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for m in re.finditer(r"ABI version (\d+)", design):
        assert int(m.group(1)) == _lib.ABI_VERSION, "DESIGN.md states ABI version %s" % m.group(1)
    # every entry point the header declares is bound (and nothing else is)
    hdr = open(os.path.join(ROOT, "include", "gstvd_hip.h")).read()
    hdr = hdr.split("#ifdef GSTVD_DIAG")[0]
    declared = set(re.findall(r"\b(gstvd_[a-z0-9_]+)\s*\(", hdr)) - {"gstvd_stream_t"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


@pytest.mark.parametrize("mode", [3])
def test_xcd_block_map_rounds_keep_whole_long_units_inside_one_round_of_32(mode):
    """Mode 3 (the fused launch's placement): an XCD's queue is a sequence of ROUNDS -- whole long-K units that together fit its 32 CUs, then
    short-K fillers for the CUs left over -- so that the tiles of a unit start together; still a placement of every tile exactly
    once, a small problem never split over XCDs, the queues balanced."""
    from gst_visdial_amd.ops import xcd_block_map, N_XCD, XCD_CUS
    shapes = _step_table()
    T = 256
    per = [((M + T - 1) // T) * ((N + T - 1) // T) for M, N, K in shapes]
    total = sum(per)
    bm = xcd_block_map(shapes, T, True, mode)
    assert len(bm) % N_XCD == 0 and sorted(t for t in bm if t >= 0) == list(range(total))
    owner, kk = [], []
    for i, n in enumerate(per):
        owner += [i] * n
        kk += [shapes[i][2]] * n
    kmax = max(kk)
    nrounds = []
    for x in range(N_XCD):
        q = [t for t in bm[x::N_XCD] if t >= 0]
        runs, start = [], 0                                          # maximal runs of long-K tiles
        for j in range(1, len(q) + 1):
            if j == len(q) or (2 * kk[q[j]] > kmax) != (2 * kk[q[j - 1]] > kmax):
                runs.append(q[start:j]); start = j
        long_runs = [r for r in runs if 2 * kk[r[0]] > kmax]
        assert all(len(r) <= XCD_CUS for r in long_runs)              # a round's long tiles fit the XCD's CUs
        assert 2 * kk[q[0]] > kmax                                    # the launch starts with a round, not with fillers
        # a long problem's tiles on this XCD sit in at most ceil(n / 27) rounds, each piece a run of consecutive ids
        for r in long_runs:
            ids = sorted(r)
            pieces = sum(1 for a, b in zip(ids, ids[1:]) if b != a + 1) + 1
            assert pieces <= len(set(owner[t] for t in r)) + 1
        for i in set(owner[t] for t in q):
            if per[i] <= 27:
                assert sum(1 for t in q if owner[t] == i) == per[i]   # never split over XCDs
        nrounds.append(len(long_runs))
    assert max(nrounds) - min(nrounds) <= 3

    def cost(t):
        return 35.0 + 0.76 * ((kk[t] + 31) // 32)
    loads = [sum(cost(t) for t in bm[x::N_XCD] if t >= 0) for x in range(N_XCD)]
    assert max(loads) / min(loads) < 1.03
