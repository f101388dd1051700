"""Host-side logic of the product that runs without a GPU: the row-selection driver (train_gen.forward's index
work, bit-exact against the oracle's restatement), the token-id decoding filters (bit-exact against the golden
vectors produced by the reference), the LR schedule, the flat parameter plan and the 861-key checkpoint layout."""
import json
import os
import tempfile

import pytest
import torch

from conftest import GOLDEN, load_npz
from oracle import vd_oracle as O


def _dialog_batch(B=3, rounds=4, T=12, U=6, R=5, F=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    b = dict(enc_input_ids=torch.randint(1, 50, (B, rounds, 1, T), generator=g),
             enc_segments=torch.randint(0, 2, (B, rounds, 1, T), generator=g),
             enc_att_mask=torch.ones(B, rounds, 1, T),
             dec_input_ids=torch.randint(1, 50, (B, rounds, 1, U), generator=g),
             dec_att_mask=torch.ones(B, rounds, 1, U),
             dec_labels=torch.randint(1, 50, (B, rounds, 1, U), generator=g),
             enc_image_feat=torch.randn(B, R, F, generator=g), enc_image_loc=torch.rand(B, R, 5, generator=g),
             enc_image_mask=torch.ones(B, R))
    b["dec_labels"][0, 1] = 0            # rows filtered by -select_data carry all-zero labels
    b["dec_labels"][2, 3] = 0
    return b


def test_select_rows_matches_reference_semantics():
    from gst_visdial_amd.step import select_rows
    b = _dialog_batch()
    params = dict(mode="vd_train", batch_size=7)
    rows, idx = select_rows(b, params, generator=torch.Generator().manual_seed(1))
    labels_flat = b["dec_labels"].reshape(-1, b["dec_labels"].shape[-1])
    cand = O.candidate_rows(labels_flat)
    assert idx.shape == (7,) and torch.all(cand[idx] == 1)          # only rows with a non-zero label row are drawn
    # same draw as the reference's torch.multinomial on the same candidate weights and generator state
    ref_idx = torch.multinomial(cand, 7, replacement=True, generator=torch.Generator().manual_seed(1))
    assert torch.equal(idx, ref_idx)
    # the reference expands image tensors 10x on the host then gathers rows (train_gen.py:311-321,45-116)
    rounds = b["enc_input_ids"].shape[1]
    exp = {k: v for k, v in b.items()}
    for k in ("enc_image_feat", "enc_image_loc", "enc_image_mask"):
        v = b[k]
        exp[k] = v.unsqueeze(1).unsqueeze(1).expand(v.shape[0], rounds, 1, *v.shape[1:]).contiguous()
    ref = O.flatten_and_gather(exp, idx)
    for k in ("enc_input_ids", "enc_segments", "enc_att_mask", "dec_input_ids", "dec_att_mask", "dec_labels",
              "enc_image_feat", "enc_image_loc", "enc_image_mask"):
        assert torch.equal(rows[k], ref[k]), k
    rows2, _ = select_rows(exp, params, sample_indices=idx)          # expanded (reference) layout gives the same rows
    for k in rows:
        assert torch.equal(rows[k], rows2[k]), k


def test_select_rows_eval_keeps_every_row_in_order():
    from gst_visdial_amd.step import select_rows
    b = _dialog_batch()
    rows, idx = select_rows(b, dict(mode="vd_eval_val", batch_size=7))
    assert torch.equal(idx, torch.arange(12)) and "dec_labels" not in rows
    assert torch.equal(rows["enc_input_ids"], b["enc_input_ids"].reshape(12, -1))


def test_decoding_filters_bit_exact(utils_golden):
    from gst_visdial_amd import decoding as D
    u = utils_golden

    def fin(t):
        return torch.where(torch.isinf(t), torch.full_like(t, -1e30), t)

    assert torch.equal(fin(D.batch_top_k_top_p_sampling(u["logits"].clone(), top_k=5)), u["topk5"])
    assert torch.equal(fin(D.batch_top_k_top_p_sampling(u["logits"].clone(), top_k=0, top_p=0.6)), u["topp06"])
    assert torch.equal(fin(D.batch_ngram_blocking(u["logits"].clone(), u["hist"], u["dec"], 3)), u["ngram3"])
    assert torch.equal(fin(D.batch_ngram_blocking(u["logits"].clone(), u["hist"], u["dec"], 2)), u["ngram2"])
    s = torch.tensor([[5, 102, 7, 102], [1, 2, 3, 4], [102, 9, 9, 9]])
    assert D.pad_after_eos(s, 102, 0).tolist() == [[5, 102, 0, 0], [1, 2, 3, 4], [102, 0, 0, 0]]


def test_lr_schedule_matches_reference(utils_golden):
    from gst_visdial_amd.optim import warmup_linear_nonzero
    lrs = torch.tensor([warmup_linear_nonzero(s, 10, 40, 2e-5) for s in range(45)], dtype=torch.float64)
    assert (lrs - utils_golden["lrs"].double()).abs().max().item() < 1e-12


def _tiny_model():
    from gst_visdial_amd.modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    cfg = json.load(open(os.path.join(GOLDEN, "tiny_cfg.json")))
    d = tempfile.mkdtemp()
    json.dump(cfg["enc"], open(d + "/e.json", "w"))
    json.dump(cfg["dec"], open(d + "/d.json", "w"))
    params = dict(model_enc_config=d + "/e.json", model_dec_config=d + "/d.json", gpu_ids=[0], model="enc_dec_a",
                  mode="vd_train", batch_size=3)
    enc, dec = VisualDialogEncoder(params), VisualDialogDecoder(params)
    return EncoderDecoderModel(params, enc, dec), enc, dec


def test_state_dict_layout_and_aliasing(tiny_state):
    model, enc, dec = _tiny_model()
    assert set(model.state_dict().keys()) == set(tiny_state.keys())
    w_before = dec.decoder.lm_head.decoder.weight
    assert w_before is dec.decoder.bert.embeddings.word_embeddings.weight        # tied at construction
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings            # train_gen.py:293
    assert dec.decoder.lm_head.decoder.weight is w_before                        # ... and untied after the swap
    assert dec.decoder.lm_head.decoder.weight is not enc.bert_pretrained.bert.embeddings.word_embeddings.weight
    assert dec.decoder.lm_head.bias is dec.decoder.lm_head.decoder.bias
    assert enc.bert_pretrained.cls.predictions.decoder.weight is enc.bert_pretrained.bert.embeddings.word_embeddings.weight
    model.load_state_dict(tiny_state, strict=True)
    assert model.decoder.config.eos_token_id == 102 and model.decoder.config.pad_token_id == 0


def test_full_config_has_861_keys():
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    from gst_visdial_amd import modules as M
    d = tempfile.mkdtemp()
    e, c = bert_base_enc_config(), bert_base_dec_config()
    for k in ("hidden_size", "v_hidden_size", "bi_hidden_size"):          # same tree, thin tensors: fast on CPU
        e[k] = 64
    e.update(intermediate_size=64, v_intermediate_size=64, num_attention_heads=2, v_num_attention_heads=2,
             bi_num_attention_heads=2, vocab_size=128, v_feature_size=32)
    c.update(hidden_size=64, intermediate_size=64, num_attention_heads=2, vocab_size=128)
    json.dump(e, open(d + "/e.json", "w")); json.dump(c, open(d + "/d.json", "w"))
    params = dict(model_enc_config=d + "/e.json", model_dec_config=d + "/d.json", gpu_ids=[0], model="enc_dec_a", mode="vd_train")
    enc, dec = M.VisualDialogEncoder(params), M.VisualDialogDecoder(params)
    model = M.EncoderDecoderModel(params, enc, dec)
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings
    assert len(model.state_dict()) == 861


def test_flat_plan_groups_and_dead_params(tiny_state):
    from gst_visdial_amd.engine import FlatParams
    model, enc, dec = _tiny_model()
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings
    fp = FlatParams(model, "bf16")
    nog = set(json.load(open(os.path.join(GOLDEN, "tiny_nograd_keys.json"))))
    names = {id(p): n for n, p in model.named_parameters()}
    assert set(names[id(p)] for p in fp.dead) == nog                             # the 42 no-grad tensors of the reference
    off, shape = fp.slots["t0.qkv.w"]
    lay = enc.bert_pretrained.bert.encoder.layer[0].attention.self
    assert shape == (3 * 64, 64)
    assert [fp.placed[id(p)] for p in (lay.query.weight, lay.key.weight, lay.value.weight)] == [off, off + 4096, off + 8192]
    assert fp.slots["dec.ckv.w"][1] == (2 * 2 * 64, 64) and fp.slots["lm.w"][1] == (320, 64)
    offs = sorted(o for _, o in fp.items)
    assert all(o % 4 == 0 for o in offs) and fp.n_live % 64 == 0
    marks = [fp.marks[k] for k in [("t", 0), ("c", 0), "vlf", "dec", ("d", 0), "lm"]]
    assert marks == sorted(marks)                                                # forward order == flat order


def test_product_refuses_cpu():
    import pytest
    from gst_visdial_amd._lib import GstvdError
    model, enc, dec = _tiny_model()
    g = load_npz("tiny_train.npz")
    with pytest.raises(GstvdError):
        model(enc_image_features=g["in::enc_image_features"], enc_image_spatials=g["in::enc_image_spatials"],
              enc_image_mask=g["in::enc_image_mask"], enc_input_ids=g["in::enc_input_ids"], enc_segments=g["in::enc_segments"],
              enc_attention_mask=g["in::enc_attention_mask"], dec_input_ids=g["in::dec_input_ids"],
              dec_attention_mask=g["in::dec_attention_mask"], dec_labels=g["in::dec_labels"])


def test_metrics_bit_exact(utils_golden):
    from gst_visdial_amd import metrics as M
    u = utils_golden
    assert torch.equal(M.scores_to_ranks(u["scores"]), u["ranks"])
    sp = M.SparseGTMetrics()
    sp.observe(u["scores"], u["gt"])
    m = sp.retrieve()
    got = torch.tensor([m[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")], dtype=torch.float64)
    assert (got - u["sparse"].double()).abs().max().item() < 1e-6
    nd = M.NDCG()
    nd.observe(u["scores"][:, 0], u["relevance"])
    assert abs(nd.retrieve()["ndcg"] - float(u["ndcg"][0])) < 1e-6


def test_checkpoint_round_trip_reference_layout(tiny_state, tmp_path):
    from gst_visdial_amd.checkpoint import save_checkpoint, load_checkpoint
    model, enc, dec = _tiny_model()
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings
    model.load_state_dict(tiny_state, strict=True)
    p = str(tmp_path / "vd_train_0_1.ckpt")
    save_checkpoint(p, model, None, iter_id=77)
    ck = torch.load(p)
    assert set(ck.keys()) == {"model_state_dict", "scheduler_state_dict", "optimizer_state_dict", "iter_id"}      # train_gen.py:346-351
    assert set(ck["model_state_dict"].keys()) == set(tiny_state.keys())
    # a checkpoint written in the reference's format (here: the golden state dict) resumes with -continue semantics
    ref = str(tmp_path / "ref.ckpt")
    torch.save({"model_state_dict": tiny_state, "iter_id": 5, "optimizer_state_dict": {}, "scheduler_state_dict": {}}, ref)
    m2, e2, d2 = _tiny_model()
    d2.decoder.bert.embeddings = e2.bert_pretrained.bert.embeddings
    assert load_checkpoint(ref, m2, cont=True) == 5
    for k, v in m2.state_dict().items():
        assert torch.equal(v, tiny_state[k]), k
    # encoder-only ingestion (train_gen.py:278-289): decoder / fusion keep their init
    m3, e3, d3 = _tiny_model()
    d3.decoder.bert.embeddings = e3.bert_pretrained.bert.embeddings
    before = m3.vlfusion.fc_l.weight.clone()
    load_checkpoint(ref, m3, cont=False)
    assert torch.equal(m3.vlfusion.fc_l.weight, before)
    assert torch.equal(m3.encoder.bert_pretrained.bert.encoder.layer[0].attention.self.query.weight,
                       tiny_state["encoder.bert_pretrained.bert.encoder.layer.0.attention.self.query.weight"])


def test_product_package_never_imports_oracle():
    """oracle/ is test infrastructure: no module of the shipped package may import it (only tests/, smoke(), bench's
    cpu_baseline leg do)."""
    import re
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gst_visdial_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+\s*oracle)", re.M)
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            with open(os.path.join(pkg, fn)) as f:
                assert not pat.search(f.read()), fn


def test_bench_synthetic_rows_follow_the_survey_generator():
    """the synthetic batch bench.py times has the structure train_gen.forward hands to the model."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    B, T, R, U, F, V = 6, 64, 37, 25, 32, 30522
    b = bench.synthetic_rows(B, T, R, U, F, V, 1234, "cpu")
    b2 = bench.synthetic_rows(B, T, R, U, F, V, 1234, "cpu")
    assert all(torch.equal(b[k], b2[k]) for k in b)                              # seeded
    ids, att, seg = b["enc_input_ids"], b["enc_attention_mask"], b["enc_segments"]
    assert ids.shape == (B, T) and (ids[:, 0] == 101).all()
    assert torch.equal(att, (ids != 0).float())
    lens = att.sum(1)
    assert (lens >= 0.6 * T - 1).all() and (lens <= T).all()
    assert ((ids == 0) | (ids == 101) | (ids == 102) | ((ids >= 1000) & (ids < 30000))).all()
    assert set(seg.unique().tolist()) <= {0, 1} and (seg[ids == 0] == 0).all()
    assert b["enc_image_features"].shape == (B, R, F) and (b["enc_image_features"] >= 0).all()
    assert torch.allclose(b["enc_image_features"][:, 0], b["enc_image_features"][:, 1:].mean(1), atol=1e-6)
    assert torch.equal(b["enc_image_spatials"][:, 0], torch.tensor([0., 0., 1., 1., 1.]).expand(B, 5))
    dec, lab, dmask = b["dec_input_ids"], b["dec_labels"], b["dec_attention_mask"]
    assert (dec[:, 0] == 101).all() and dec.shape == lab.shape == (B, U)
    for r in range(B):                                                           # labels = decoder ids shifted left + [SEP]
        n = int((lab[r] != 0).sum())
        assert lab[r, n - 1] == 102 and (lab[r, n:] == 0).all()
        assert torch.equal(dec[r, 1:n], lab[r, :n - 1]) and (dec[r, n:] == 0).all()
        assert dmask[r].sum() >= n


def _append_reference_semantics(ctx, ctx_len, new, sep, seg=None):
    """Per-row restatement of what generate.py:145-158 / 203-218 do (slice assignment; on a size mismatch a lone [SEP])."""
    B, T = ctx.shape
    bad = []
    lens = (new != 0).sum(-1)
    for b in range(B):
        n = int(lens[b]); start = int(ctx_len[b]); end = start + n
        if end <= T:
            ctx[b, start:end] = new[b, :n]
        else:
            ctx[b, start:start + 1] = torch.tensor([sep])
            n = 1; end = start + 1; bad.append(b)
        if seg is not None:
            seg[b, start:end] = 1
        ctx_len[b] += n
    return bad


def test_append_to_context_matches_per_row_reference_semantics():
    from gst_visdial_amd.generate import append_to_context
    g = torch.Generator().manual_seed(3)
    B, T, U = 9, 40, 18
    for trial in range(20):
        ctx_len = torch.randint(5, T - 1, (B,), generator=g)
        ctx = torch.randint(5, 300, (B, T), generator=g) * (torch.arange(T)[None] < ctx_len[:, None])
        n = torch.randint(0, U + 1, (B,), generator=g)
        new = torch.randint(5, 300, (B, U), generator=g) * (torch.arange(U)[None] < n[:, None])
        seg = torch.zeros(B, T, dtype=torch.long)
        c1, l1, s1 = ctx.clone(), ctx_len.clone(), seg.clone()
        c2, l2, s2 = ctx.clone(), ctx_len.clone(), seg.clone()
        bad_ref = _append_reference_semantics(c1, l1, new, 102, s1)
        n_eff, bad = append_to_context(c2, l2, new, 102, segments=s2, segment_value=1)
        assert torch.equal(c1, c2) and torch.equal(l1, l2) and torch.equal(s1, s2)
        assert bad.tolist() == bad_ref
        assert torch.equal(n_eff, l2 - ctx_len)


def test_vectorised_ngram_blocking_equals_the_reference_loops():
    from gst_visdial_amd.decoding import batch_ngram_blocking, _ngram_blocking_loop
    g = torch.Generator().manual_seed(5)
    V = 40                                                   # small vocabulary: plenty of repeated n-grams
    for trial in range(60):
        B, T = 4, int(torch.randint(3, 30, (1,), generator=g))
        n = int(torch.randint(0, 6, (1,), generator=g))
        cur = int(torch.randint(1, 12, (1,), generator=g))
        hist = torch.randint(0, V, (B, T), generator=g)
        hist[torch.rand(B, T, generator=g) < 0.1] = 102      # sprinkle special tokens
        dec = torch.randint(0, V, (B, cur), generator=g)
        if T >= 3 and cur >= 2:                              # force some real matches
            dec[0, -2:] = hist[0, 1:3]
        logits = torch.randn(B, V + 70, generator=g)
        a = batch_ngram_blocking(logits.clone(), hist, dec, ngram_size=n)
        b = _ngram_blocking_loop(logits.clone(), hist, dec, ngram_size=n)
        assert torch.equal(a, b), (trial, n, cur, T)


def test_reference_optimizer_param_order_matches_the_reference_run():
    """optim.reference_param_index == the order (and the set of tensors that ever get optimizer state) of the reference's own
    per-tensor param groups (tests/golden/tiny_trainer.json, written by oracle/make_golden_r2.py from the real modules)."""
    from gst_visdial_amd.optim import reference_param_index
    from gst_visdial_amd.engine import FlatParams
    meta = json.load(open(os.path.join(GOLDEN, "tiny_trainer.json")))
    model, enc, dec = _tiny_model()
    with pytest.raises(NotImplementedError):
        reference_param_index(model)                                             # defined for the aliased (train_gen.py) set-up
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings            # train_gen.py:293
    idx = reference_param_index(model)
    assert [n for n, _ in idx] == meta["param_names"]
    assert [list(p.shape) if p is not None else meta["shapes"][i] for i, (_, p) in enumerate(idx)] == meta["shapes"]
    fp = FlatParams(model, "fp32")
    live = set(id(p) for p in fp.live)
    assert [i for i, (_, p) in enumerate(idx) if p is not None and id(p) in live] == meta["stateful_indices"]
    assert not any(n.startswith("vlfusion") for n, _ in idx)                     # never in the reference's optimizer


def test_inverse_cdf_draw():
    from gst_visdial_amd.decoding import draw_from_uniform
    p = torch.tensor([[0.0, 0.2, 0.0, 0.8], [0.5, 0.5, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    assert draw_from_uniform(p, torch.tensor([0.1, 0.7, 0.999])).view(-1).tolist() == [1, 1, 3]
    assert draw_from_uniform(p, torch.tensor([0.3, 0.2, 1e-6])).view(-1).tolist() == [3, 0, 3]
    g = torch.Generator().manual_seed(0)
    pr = torch.softmax(torch.randn(1, 50, generator=g), -1).expand(20000, 50)
    idx = draw_from_uniform(pr, torch.rand(20000, generator=g)).view(-1)
    freq = torch.bincount(idx, minlength=50).float() / 20000
    assert (freq - pr[0]).abs().max() < 0.01                                     # it IS a draw from prob


def test_capture_guard_collects_before_and_disables_gc_during():
    import gc
    from gst_visdial_amd import graph
    seen = []
    with graph.gc_quiet():
        seen.append(gc.isenabled())
        with graph.gc_quiet():
            seen.append(gc.isenabled())
        seen.append(gc.isenabled())
    assert seen == [False, False, False] and gc.isenabled()


def test_deepcopy_of_a_model_gets_its_own_engine():
    """the engine holds its model weakly; copy.deepcopy(model) must not leave the copy's engine bound to the
    original (nor give the copy's encoder / decoder holders a third engine)."""
    import copy
    model, enc, dec = _tiny_model()
    eng = model.engine
    twin = copy.deepcopy(model)
    assert twin.engine is not eng and twin.engine.model is twin and eng.model is model
    assert twin.encoder._standalone_engine is twin.engine and twin.decoder._standalone_engine is twin.engine


def test_rank_seed_gives_every_rank_its_own_dropout_stream():
    from gst_visdial_amd import ops
    seeds = [ops.rank_seed(1234, r) for r in range(8)]
    assert seeds[0] == 1234 and len(set(seeds)) == 8 and all(0 <= s < 2 ** 63 for s in seeds)
    assert len({s & 0xffffffff for s in seeds}) == 8 and len({s >> 32 for s in seeds}) == 8     # both words the kernels hash differ


def _np_mix32(x):
    """csrc/common.h mix32 restated in numpy (uint32 wrap-around arithmetic; __umul24 = product of the low 24 bits, low 32 kept)."""
    import numpy as np
    x = x.astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xeb352d)) & m
    x ^= x >> np.uint64(15); x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0x6ca68b)) & m
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def _np_drop_hash(c, key):
    import numpy as np
    c = c.astype(np.uint64)
    t = ((c >> np.uint64(24)) * np.uint64(0x9E3779)) & np.uint64(0xFFFFFFFF)
    return _np_mix32((c ^ np.uint64(key) ^ t) & np.uint64(0xFFFFFFFF))


def test_dropout_counter_hash_has_no_structured_top_byte_collisions():
    """the 24-bit multiplies of mix32 dropped the counter's top byte -- mix32(c ^ key) == mix32((c ^ 0x01000100) ^ key)
    for EVERY c and key, i.e. exact mask duplicates at a fixed distance once a site has more than 2^24 draws.  drop_hash (the
    top byte hashed into the key first) must (a) leave the first 2^24 counters' stream unchanged, (b) have no such pair."""
    import numpy as np
    rs = np.random.RandomState(0)
    c = rs.randint(0, 2 ** 32, size=1 << 18, dtype=np.uint64).astype(np.uint32)
    for key in (0, 0x12345678, 0xdeadbeef):
        old = _np_mix32(c ^ np.uint32(key))
        assert np.array_equal(old, _np_mix32((c ^ np.uint32(0x01000100)) ^ np.uint32(key)))      # the defect, restated
        new = _np_drop_hash(c, key)
        low = c < (1 << 24)
        assert np.array_equal(new[low], old[low])                                                   # (a)
        for flip in (0x01000100, 0x02000200, 0x80008000, 0x01000000, 0xff000000, 0x81008100):
            same = (new == _np_drop_hash(c ^ np.uint32(flip), key)).mean()
            assert same < 1e-3, (hex(key), hex(flip), same)                                       # (b): unrelated draws
    # counters that differ ONLY in bits 24..31 (256 blocks of one low part): all 256 draws distinct for almost every low part
    lowpart = rs.randint(0, 1 << 24, size=4096).astype(np.uint32)
    blocks = (np.arange(256, dtype=np.uint32)[:, None] << np.uint32(24)) | lowpart[None, :]
    d = _np_drop_hash(blocks.reshape(-1), 0x9e3779b9).reshape(256, -1)
    distinct = np.array([len(np.unique(d[:, j])) for j in range(d.shape[1])])
    assert (distinct >= 255).mean() > 0.99 and distinct.min() >= 250
    # sequential counters across a 2^24 boundary keep the keep-rate and show no lag-2^24 correlation
    seq = np.arange((1 << 24) - (1 << 16), (1 << 24) + (1 << 16), dtype=np.uint32)
    keep = (_np_drop_hash(seq, 0x1234) & 0xffff) >= int(0.1 * 65536 + 0.5)
    assert abs(keep.mean() - 0.9) < 5e-3
    a = (_np_drop_hash(seq[: 1 << 16], 0x1234) & 0xffff) >= 6554
    b = (_np_drop_hash(seq[: 1 << 16] + np.uint32(1 << 24), 0x1234) & 0xffff) >= 6554
    assert abs((a == b).mean() - (0.81 + 0.01)) < 0.01


def test_fused_update_block_cover_rules():
    """optim._FuseHandle.cover(): which weight-gradient blocks the grouped launch may update itself (gstvd_gemm_grouped_adamw) --
    one whole tensor; several consecutive tensors with one (lr, wd) (the fused Q|K|V projections); a tensor followed by alignment
    padding (the LM head's padded rows); never a frozen tensor, a block that stops inside a tensor, tensors with different
    hyper-parameters, or a block that does not start a tensor."""
    from gst_visdial_amd.optim import _FuseHandle

    class T(object):
        def __init__(self, p):
            self.p = p

        def data_ptr(self):
            return self.p

    class O(object):
        pass
    opt = O()
    opt.hp = T(7 << 20)
    # q, k, v [64 x 32] back to back, a bias, padding, a frozen [64 x 32] tensor, a [60 x 32] head padded to 64 rows
    opt.seg_ends_host = [2048, 4096, 6144, 6208, 6272, 8320, 10240, 10368]
    opt.base = [(1e-3, 0.01, 1.0)] * 3 + [(1e-3, 0.0, 1.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1e-3, 0.01, 1.0), (0.0, 0.0, 0.0)]
    opt.seg_of = {0: (0, 2048), 2048: (1, 2048), 4096: (2, 2048), 6144: (3, 64), 6272: (5, 2048), 8320: (6, 1920)}
    h = _FuseHandle.__new__(_FuseHandle)
    h.opt, h._g0 = opt, 4096
    g = lambda off: 4096 + 4 * off
    hp = lambda seg: (7 << 20) + 8 * seg
    assert h.cover(g(0), 192, 32, 32) == (hp(0), (0, 2048, 4096))
    assert h.cover(g(0), 64, 32, 32) == (hp(0), (0,))
    assert h.cover(g(2048), 128, 32, 32) == (hp(1), (2048, 4096))
    assert h.cover(g(8320), 64, 32, 32) == (hp(6), (8320,)) and h.cover(g(8320), 60, 32, 32) == (hp(6), (8320,))
    for args in ((g(0), 160, 32, 32), (g(4096), 66, 32, 32), (g(6272), 64, 32, 32), (g(8320), 70, 32, 32), (g(100), 4, 32, 32),
                 (g(0), 64, 32, 48), (g(0) + 4, 64, 32, 32), (g(0), 64, 30, 30)):
        assert h.cover(*args) == (0, ()), args


def test_fused_update_remainder_block_list():
    """optim.FusedAdamW._remainder(): the 1024-element blocks the remainder pass must visit behind a weight-gradient launch that
    updated the listed weights itself -- every block that holds a live element of [lo, hi) outside those weights, no block that
    holds only fused weights / padding / frozen tensors -- and the per-segment skip flags."""
    import torch
    from gst_visdial_amd.optim import FusedAdamW
    opt = FusedAdamW.__new__(FusedAdamW)
    # layout (elements): embedding [0, 5000) | W1 [5056, 5056 + 8192) | b1 64 | pad | frozen [13376, 14400) | W2 [14400, 14400 + 4096) | b2 [18496, 18560)
    opt.seg_ends_host = [5000, 5056, 13248, 13312, 13376, 14400, 18496, 18560]
    live, dead = (1e-3, 0.01, 1.0), (0.0, 0.0, 0.0)
    opt.base = [live, dead, live, (1e-3, 0.0, 1.0), dead, dead, live, (1e-3, 0.0, 1.0)]
    opt.seg_of = {0: (0, 5000), 5056: (2, 8192), 13248: (3, 64), 13376: (5, 1024), 14400: (6, 4096), 18496: (7, 64)}
    opt.seg_end = torch.tensor(opt.seg_ends_host, dtype=torch.int64)
    opt._remainders = {}
    blocks, skip = opt._remainder(0, 18560, (5056, 14400))
    assert skip.tolist() == [0, 0, 1, 0, 0, 0, 1, 0]
    # embedding: blocks 0..4; b1 [13248, 13312): block 12; b2 [18496, 18560): block 18; the frozen tensor and the padding need none
    assert blocks.tolist() == [0, 1, 2, 3, 4, 12, 18]
    blocks, skip = opt._remainder(13248, 18560, (14400,))          # a later slice: W1 is not part of it, the embedding neither
    assert blocks.tolist() == [12, 18] and skip.tolist() == [0, 0, 0, 0, 0, 0, 1, 0]
    blocks, _ = opt._remainder(0, 13248, ())                        # nothing fused in this slice: every live block of the range
    assert blocks.tolist() == list(range(0, (13248 - 1) // 1024 + 1))     # the embedding's blocks 0..4, W1's 4..12
