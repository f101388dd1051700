"""GPU parity tests anchored on the round-2 golden vectors (oracle/make_golden_r2.py: the REAL reference run on an eval set,
under real sampling, and in train_gen.py's loop with a reference-format optimizer state).  Everything goes through the
module API / C ABI; fp32 parity mode: logits 1e-4, scores 1e-3 (a sum of <= 9 log-probabilities), ids / ranks bit-exact."""
import json
import os

import pytest
import torch

from conftest import load_npz, GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sc():
    from gst_visdial_amd import selfcheck
    return selfcheck


def maxerr(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def _eval_batch(ev, tag=""):
    return {k[len(tag) + 4:]: v.clone() for k, v in ev.items() if k.startswith(tag + "in::")}


# ------------------------------------------------------------------------------------------------ f-1: evaluate_gen
@pytest.mark.parametrize("tag", ["", "attacked::"])
def test_eval_set_scores_ranks_metrics_match_the_reference_run(tag):
    """evaluate_gen.py:45-118 end to end: gst_visdial_amd.evaluate.score_batch -> [B, rounds, options] scores equal the
    reference's expanded run; ranks bit-exact; evaluate() returns the reference's R@k / mean / MRR / NDCG.  The clean set
    takes the encode-once path; the 'attacked' set (one option row sees a perturbed context) must NOT."""
    from gst_visdial_amd import evaluate as EV
    from gst_visdial_amd.metrics import scores_to_ranks
    s = sc()
    model, params, cfg = s.build_tiny_model("fp32", DEV, mode="vd_eval_val")
    model.eval()
    ev = load_npz("tiny_evalset.npz")
    batch = _eval_batch(ev, tag)
    calls = []
    orig = model.score_candidates
    model.score_candidates = lambda *a: (calls.append(a[-1]), orig(*a))[1]
    scores = EV.score_batch(model, batch, torch.device(DEV)).cpu()
    assert set(calls) == ({5} if tag == "" else {1})              # encode-once only when the contexts really are shared
    assert maxerr(scores, ev[tag + "scores"]) < 1e-3
    assert torch.equal(scores_to_ranks(scores), ev[tag + "ranks"])
    p = dict(params, device=torch.device(DEV), vd_version="1.0")
    _, metrics = EV.evaluate(model, [batch], p, mode="vd_eval_val")
    got = torch.tensor([metrics[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")], dtype=torch.float64)
    assert maxerr(got, ev[tag + "sparse"]) < 1e-6
    assert abs(metrics["ndcg"] - ev[tag + "ndcg"].item()) < 1e-6
    # test-split path (evaluate_gen.py:119-133): one round per dialog, ranks of its options as a json list
    one = {k: (v[:, 1:2] if v.dim() == 4 else v) for k, v in batch.items()}
    ranks_json, _ = EV.evaluate(model, [dict(one, image_id=torch.tensor([7, 8]))], p, mode="vd_eval_test")
    assert len(ranks_json) == 2 and ranks_json[0]["image_id"] == 7 and ranks_json[1]["round_id"] == 3
    assert ranks_json[0]["ranks"] == ev[tag + "ranks"][0, 1].tolist()


def test_score_candidates_matches_reference_scores_directly():
    """Engine.score_candidates (one encoder pass per round, shared cross-K/V, fused log-softmax + gather) against the
    reference's own numbers -- replaces round 1's engine-vs-itself comparison as the anchor."""
    s = sc()
    model, params, cfg = s.build_tiny_model("fp32", DEV, mode="vd_eval_val")
    model.eval()
    ev = load_npz("tiny_evalset.npz")
    b = _eval_batch(ev)
    B, NR, G, T = b["enc_input_ids"].shape
    dial = torch.arange(B).repeat_interleave(NR)
    d = lambda t: t.to(DEV)
    sc_ = model.score_candidates(d(b["enc_image_feat"][dial]), d(b["enc_image_loc"][dial]), d(b["enc_image_mask"][dial]),
                                 d(b["enc_input_ids"][:, :, 0].reshape(B * NR, T)), d(b["enc_segments"][:, :, 0].reshape(B * NR, T)),
                                 d(b["enc_att_mask"][:, :, 0].reshape(B * NR, T)), d(b["dec_input_ids"].reshape(B * NR * G, -1)),
                                 d(b["dec_att_mask"].reshape(B * NR * G, -1)), G)
    assert maxerr(sc_.view(B, NR, G), ev["scores"]) < 1e-3


# ------------------------------------------------------------------------------------------------ f-2: sampling decode
def _decode_kw(s, g):
    kw = s.golden_batch(g, DEV)
    kw["dec_input_ids"] = torch.full((kw["enc_input_ids"].shape[0], 1), 101, dtype=torch.long, device=DEV)
    kw["dec_labels"] = None
    return kw


def test_sampled_ids_equal_the_reference_under_the_same_uniforms():
    """Real sampling (temperature 1.3, top-k 40, bigram ban): ids drawn on the device by inverse CDF from caller-supplied
    uniforms are bit-equal to the reference's run under the same uniforms -- eager issue and hipGraph replay."""
    s = sc()
    model, params, cfg = s.build_tiny_model("fp32", DEV, mode="vd_gen_val")
    model.eval()
    g, sm = load_npz("tiny_train.npz"), load_npz("tiny_sampled.npz")
    args = dict(temperature=float(sm["temperature"]), top_k=int(sm["top_k"]), top_p=0.0, ngram_blocking_size=2,
                uniforms=sm["uniforms"].to(DEV))
    a0 = model(**args, **_decode_kw(s, g))                    # eager (then captures the session)
    a1 = model(**args, **_decode_kw(s, g))                    # replay
    assert torch.equal(a0.cpu(), sm["sequence"]) and torch.equal(a1.cpu(), sm["sequence"])
    model.engine.close()


def test_bf16_decode_step_with_folded_layernorms_matches_the_unfolded_plan_and_fp32():
    """The bf16 decode step folds every LayerNorm into the Linear that reads it (gstvd_gemv_ln) and runs the single-query
    attention kernel: the logits after a 10-token prefix (no sampling involved: max_seq_len = 1) equal those of the plan
    with separate LayerNorm launches within bf16 rounding, and both sit at bf16 distance from the fp32-mode engine's."""
    s = sc()
    g = load_npz("tiny_train.npz")
    gen = torch.Generator().manual_seed(3)
    outs = {}
    for name, prec, fuse in (("folded", "bf16", True), ("separate", "bf16", False), ("fp32", "fp32", True)):
        model, params, cfg = s.build_tiny_model(prec, DEV, mode="vd_gen_val")
        params["amd_decode_fuse_ln"] = fuse
        params["amd_decode_graph"] = False
        model.eval()
        kw = _decode_kw(s, g)
        Bn = kw["enc_input_ids"].shape[0]
        prefix = torch.randint(104, 300, (Bn, 10), generator=torch.Generator().manual_seed(3)).to(DEV)
        prefix[:, 0] = 101
        seq = model.engine.sample(kw["enc_image_features"], kw["enc_image_spatials"], kw["enc_image_mask"], kw["enc_input_ids"],
                                  kw["enc_segments"], kw["enc_attention_mask"], prefix, temperature=1.0, top_k=1, max_seq_len=1)
        assert seq.shape == (Bn, 1)
        outs[name] = model.engine.last["decode_logits"].clone()
        model.engine.close()
    scale = outs["fp32"].abs().max().item()
    assert (outs["folded"] - outs["separate"]).abs().max().item() <= 2e-2 * max(1.0, scale)
    assert (outs["folded"] - outs["fp32"]).abs().max().item() <= 0.1 * max(1.0, scale)
    assert (outs["separate"] - outs["fp32"]).abs().max().item() <= 0.1 * max(1.0, scale)


def test_top_p_decode_runs_in_the_fused_kernel_and_equals_the_torch_filters():
    """top-p (not used by the reference's scripts, generate.py:138-141; utils/decoding_utils.py:22-34) runs inside the sampling
    kernel since ABI 6: the decode is captured like any other, and its ids are those of the torch-op form of the filters (the
    path a vocabulary beyond the kernel's range still takes), step by step under the same uniforms."""
    from gst_visdial_amd.engine import Engine
    s = sc()
    model, params, cfg = s.build_tiny_model("fp32", DEV, mode="vd_gen_val")
    model.eval()
    g, sm = load_npz("tiny_train.npz"), load_npz("tiny_sampled.npz")
    u = sm["uniforms"].to(DEV)
    args = dict(temperature=1.1, top_k=12, top_p=0.8, ngram_blocking_size=2, uniforms=u)
    a0 = model(**args, **_decode_kw(s, g))
    a1 = model(**args, **_decode_kw(s, g))
    assert len(model.engine._decode_sessions) == 1 and torch.equal(a0, a1)
    keep = Engine.__dict__["_fused_sampling"]                                  # (the staticmethod object itself)
    try:
        Engine._fused_sampling = staticmethod(lambda P, vocab: False)          # the torch-op filters, eagerly
        e0 = model(**args, **_decode_kw(s, g))
        w0 = model(**dict(args, top_k=0, top_p=0.5), **_decode_kw(s, g))
    finally:
        Engine._fused_sampling = keep
    assert torch.equal(a0, e0)
    k0 = model(**dict(args, top_k=0, top_p=0.5), **_decode_kw(s, g))          # top-p alone (no top-k in front)
    assert torch.equal(k0, w0)
    b0 = model(**dict(args, top_p=0.0), **_decode_kw(s, g))
    V = model.decoder.config.vocab_size
    assert b0.shape == a0.shape and ((a0 >= 0) & (a0 < V)).all() and ((b0 >= 0) & (b0 < V)).all()
    model.engine.close()


def test_decode_session_follows_parameter_updates():
    """a captured decode session must not keep reading stale (bf16 shadow) weights after load_state_dict."""
    s = sc()
    g = load_npz("tiny_train.npz")
    sm = load_npz("tiny_sampled.npz")
    args = dict(temperature=1.3, top_k=40, top_p=0.0, ngram_blocking_size=2, uniforms=sm["uniforms"].to(DEV))
    for prec in ("bf16", "fp32"):
        model, params, cfg = s.build_tiny_model(prec, DEV, mode="vd_gen_val")
        model.eval()
        model(**args, **_decode_kw(s, g)); model(**args, **_decode_kw(s, g))          # session captured and replayed
        sd = {k: (v + 0.05 * torch.randn(v.shape, generator=torch.Generator().manual_seed(len(k))) if v.dtype.is_floating_point else v)
              for k, v in load_npz("tiny_state.npz").items()}
        model.load_state_dict(sd)
        got = model(**args, **_decode_kw(s, g))                                         # replay with the NEW weights
        fresh, fp, _ = s.build_tiny_model(prec, DEV, mode="vd_gen_val")
        fp["amd_decode_graph"] = False
        fresh.load_state_dict(sd)
        fresh.eval()
        want = fresh(**args, **_decode_kw(s, g))
        assert torch.equal(got, want), prec
        model.engine.close()


def test_perplexity_rescore_reuses_the_decode_state_and_matches_the_oracle():
    """generate.py:183-211 ("ppl trick"): Engine.rescore_sampled scores the sampled answer against the encoder states and
    cross-attention K/V the decode call left behind -- same per-token losses as the full re-run (encoder + decoder again)
    and as the oracle, without running the encoder (checked by counting encoder-sized GEMM launches)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import vd_oracle as O
    from gst_visdial_amd.generate import answer_perplexity
    from gst_visdial_amd import ops
    s = sc()
    g = load_npz("tiny_train.npz")
    for graph in (False, True):
        model, params, cfg = s.build_tiny_model("fp32", DEV, mode="cc12m_gen")
        params["amd_decode_graph"] = graph
        model.eval()
        kw = _decode_kw(s, g)
        enc_kw = {k: v for k, v in kw.items() if k.startswith("enc_")}
        torch.manual_seed(5)
        if graph:
            model(temperature=0.7, top_k=7, **kw)                                       # first call captures
        torch.manual_seed(5)
        ans = model(temperature=0.7, top_k=7, **kw)
        sampled = ans.clone()
        with ops.Profiler() as prof:
            ppl_fast, len_fast = answer_perplexity(model, enc_kw, ans, reuse_decode_state=True)
        launches = prof.summary(by_shape=True)
        n_attn = sum(v["launches"] for k, v in launches.items() if k.startswith("attn_fwd"))
        assert n_attn == 2 * cfg["dec"]["num_hidden_layers"]                             # decoder self + cross only: no encoder pass
        ans2 = sampled.clone()
        ppl_full, len_full = answer_perplexity(model, enc_kw, ans2, reuse_decode_state=False)
        assert torch.equal(ans, ans2) and torch.equal(len_fast, len_full)                # same in-place [SEP] -> [PAD]
        assert maxerr(ppl_fast, ppl_full) <= 1e-4 * ppl_full.max().item()
        cpu = {k: v.cpu() for k, v in enc_kw.items()}
        cpu.update(dec_input_ids=sampled.cpu().clone(), dec_attention_mask=(sampled != 0).float().cpu())
        out = O.model_forward(load_npz("tiny_state.npz"), cfg["enc"], cfg["dec"], cpu, loss_reduction=False)
        B = sampled.shape[0]
        ref_len = (cpu["dec_input_ids"] != 0).sum(-1)
        ref_ppl = torch.exp(out["loss"].reshape(B, -1).sum(-1) / ref_len)
        assert maxerr(ppl_fast, ref_ppl) <= 1e-3 * ref_ppl.max().item()
        with pytest.raises(Exception):                                                   # the state is single use
            model.engine.rescore_sampled(sampled.clone())
        model.engine.close()


# ------------------------------------------------------------------------------------------------ f-4: trainer / resume
def _trainer_setup(s, meta, state=None):
    from gst_visdial_amd.optim import FusedAdamW
    model, params, cfg = s.build_tiny_model("fp32", DEV)
    if state is not None:
        model.load_state_dict(state, strict=True)
    model.eval()                                                  # dropout off, as in the golden run
    opt = FusedAdamW(model, lr=meta["lr"], image_lr=meta["image_lr"], language_weights=meta["language_weights"],
                     warmup_steps=meta["warmup_steps"], t_total=meta["t_total"])
    return model, opt


def _train_iters(model, opt, kw, first, last):
    losses = []
    for it in range(first, last + 1):                             # train_gen.py:323-329
        loss, _ = model(**kw)
        loss.backward()
        if it > 0:
            opt.step()
            opt.zero_grad()
        opt.scheduler_step()
        losses.append(loss.item())
    return losses


def _rel_state_err(model, tr, prefix):
    got = {k: t.detach().float().cpu() for k, t in model.state_dict().items()}
    worst = 0.0
    for k, v in got.items():
        ref = tr[prefix + k]
        worst = max(worst, (v - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6))
    return worst


def test_train_loop_follows_the_reference_run_including_iteration_zero():
    """Six iterations of train_gen.py's loop on the engine == the reference model's own run: iteration 0 neither steps nor
    zeroes (iteration 1 steps on the accumulated gradient), per-tensor lr (language vs image weights) and decay, warm-up
    schedule, and the VLFusion projections stay untouched (they are in none of the reference's param groups)."""
    s = sc()
    tr = load_npz("tiny_trainer.npz")
    meta = json.load(open(os.path.join(GOLDEN, "tiny_trainer.json")))
    model, opt = _trainer_setup(s, meta)
    kw = s.golden_batch(load_npz("tiny_train.npz"), DEV)
    losses = _train_iters(model, opt, kw, 0, 5)
    for a, b in zip(losses, tr["losses"].tolist()):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (losses, tr["losses"])
    assert losses[0] == losses[1]
    assert _rel_state_err(model, tr, "state5::") < 2e-3
    assert torch.equal(model.vlfusion.fc_v.weight.detach().cpu(), load_npz("tiny_state.npz")["vlfusion.fc_v.weight"])


def test_resume_from_a_reference_format_checkpoint(tmp_path):
    """`-continue` (train_gen.py:254-276) from a checkpoint in the REFERENCE's layout -- model state after iteration 3 and
    the reference optimizer's per-tensor state_dict() -- loaded BEFORE the first forward: iterations 4-5 reproduce the
    reference run (moments, bias-correction step and schedule position all restored)."""
    from gst_visdial_amd import checkpoint as CK
    s = sc()
    tr = load_npz("tiny_trainer.npz")
    meta = json.load(open(os.path.join(GOLDEN, "tiny_trainer.json")))
    state, groups = {}, []
    for i in meta["stateful_indices"]:
        state[i] = dict(step=int(tr["opt3::%d::step" % i]), exp_avg=tr["opt3::%d::exp_avg" % i], exp_avg_sq=tr["opt3::%d::exp_avg_sq" % i])
    for i, gmeta in enumerate(meta["groups_iter3"]):
        groups.append(dict(gmeta, betas=(0.9, 0.999), eps=1e-6, correct_bias=True))
    ck = {"model_state_dict": {k[8:]: v for k, v in tr.items() if k.startswith("state3::")},
          "optimizer_state_dict": {"state": state, "param_groups": groups},
          "scheduler_state_dict": dict(meta["scheduler_iter3"]), "iter_id": 3}
    path = str(tmp_path / "vd_train_ref_3.ckpt")
    torch.save(ck, path)
    model, opt = _trainer_setup(s, meta)
    it = CK.load_checkpoint(path, model, opt, cont=True)
    assert it == 3 and opt.opt_step == 3 and opt.sched_step == meta["scheduler_iter3"]["last_epoch"] == 4
    kw = s.golden_batch(load_npz("tiny_train.npz"), DEV)
    losses = _train_iters(model, opt, kw, 4, 5)
    for a, b in zip(losses, tr["losses"].tolist()[4:]):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (losses, tr["losses"])
    assert _rel_state_err(model, tr, "state5::") < 2e-3
    # and the other direction: what we export after iteration 5 has the reference's layout (indices, orphan slots, steps)
    osd = opt.export_reference_state()
    assert len(osd["param_groups"]) == len(meta["param_names"])
    assert sorted(osd["state"].keys()) == meta["stateful_indices"]
    assert all(v["step"] == 5 for v in osd["state"].values())
    i198 = meta["param_names"].index("decoder.decoder.bert.embeddings.word_embeddings.weight")
    assert osd["state"][i198]["exp_avg"].shape == model.decoder.decoder.lm_head.decoder.weight.shape


@pytest.mark.parametrize("fmt", ["flat", "reference"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_save_resume_round_trip_equals_uninterrupted_training(tmp_path, fmt, prec):
    """train k steps, save, load into a FRESH model + optimizer (before any forward), and step k+1 must
    equal the uninterrupted run (moments, device step counter, schedule position, bf16 shadow) -- fp32: identical losses,
    parameters equal up to the summation order of the embedding-gradient atomics (1e-9), in both on-disk layouts."""
    from gst_visdial_amd import checkpoint as CK
    from gst_visdial_amd.optim import FusedAdamW
    s = sc()
    kw = s.golden_batch(load_npz("tiny_train.npz"), DEV)

    def fresh():
        model, params, cfg = s.build_tiny_model(prec, DEV, seed=2)
        model.eval()
        return model, FusedAdamW(model, lr=2e-3, warmup_steps=2, t_total=40)

    def steps(model, opt, n):
        out = []
        for _ in range(n):
            loss, _ = model(**kw)
            loss.backward()
            opt.step(); opt.zero_grad(); opt.scheduler_step()
            out.append(loss.item())
        return out

    m0, o0 = fresh()
    steps(m0, o0, 3)
    path = str(tmp_path / "ck.ckpt")
    CK.save_checkpoint(path, m0, o0, iter_id=3, reference_format=(fmt == "reference"))
    cont = steps(m0, o0, 2)                                       # the uninterrupted run
    m1, o1 = fresh()
    assert CK.load_checkpoint(path, m1, o1, cont=True) == 3       # before the first forward
    assert o1.opt_step == 3 and o1.sched_step == 3
    res = steps(m1, o1, 2)
    assert float(o1.step_dev.item()) == 5.0
    if prec == "fp32":
        for a, b in zip(res, cont):          # not bit-equal across runs: the embedding-gradient scatter adds in arrival order
            assert abs(a - b) < 2e-6 * max(1.0, abs(b)), (res, cont)
        assert maxerr(m1.engine.flat.P, m0.engine.flat.P) < 1e-6
    else:
        for a, b in zip(res, cont):
            assert abs(a - b) < 5e-3 * max(1.0, abs(b)), (res, cont)


# ------------------------------------------------------------------------------------------------ f-3: host input path
def test_pinned_double_buffered_staging_and_prefetch():
    """step.PinnedStager / step.prefetch (replacing train_gen.py:102-116's pageable copies and :311-321's 10x expansion): the
    rows reach persistent device buffers through rotating pinned slots, the host half one batch ahead of the compute; five
    different batches through two slots give exactly the losses of the plain path (no slot is overwritten while in use)."""
    from gst_visdial_amd import step
    s = sc()
    model, params, cfg = s.build_tiny_model("fp32", DEV)
    model.eval()
    g = load_npz("tiny_train.npz")
    b = {k[4:]: v.clone() for k, v in g.items() if k.startswith("in::")}

    def dialog_batch(shift):
        order = torch.tensor([0, 1, 2, 2, 0, 1])
        dlg = lambda x: x[order].reshape((2, 3, 1) + tuple(x.shape[1:]))
        ids = b["enc_input_ids"]
        ids = torch.where(ids > 110, (ids - 111 + shift) % 200 + 111, ids)
        return dict(enc_input_ids=dlg(ids), enc_segments=dlg(b["enc_segments"]), enc_att_mask=dlg(b["enc_attention_mask"]),
                    dec_input_ids=dlg(b["dec_input_ids"]), dec_att_mask=dlg(b["dec_attention_mask"]), dec_labels=dlg(b["dec_labels"]),
                    enc_image_feat=b["enc_image_features"][:2] * (1.0 + 0.1 * shift), enc_image_loc=b["enc_image_spatials"][:2],
                    enc_image_mask=b["enc_image_mask"][:2])          # image tensors UNEXPANDED: one per dialog

    p = dict(params, mode="vd_train", batch_size=4, device=torch.device(DEV))
    batches = [dialog_batch(k) for k in range(5)]
    with torch.no_grad():
        plain = [step.forward(model, bt, p, generator=torch.Generator().manual_seed(50 + i))[0].item() for i, bt in enumerate(batches)]
        st = step.PinnedStager(DEV, depth=2)
        staged = [step.forward(model, bt, p, generator=torch.Generator().manual_seed(50 + i), stager=st)[0].item()
                  for i, bt in enumerate(batches)]
        assert staged == plain
        gens = iter(torch.Generator().manual_seed(50 + i) for i in range(5))

        class Loader(object):                      # a generator per batch, like the sequential runs above
            def __iter__(self):
                return iter(batches)
        st2 = step.PinnedStager(DEV, depth=2)
        pre = []
        it = iter(batches)
        pending = None
        for i, bt in enumerate(batches):           # prefetch() by hand so each batch keeps its own generator
            rows, _ = step.select_rows(bt, p, None, torch.Generator().manual_seed(50 + i))
            h = st2.put(rows)
            if pending is not None:
                pre.append(step.forward_rows(model, st2.get(pending), p)[0].item())
            pending = h
        pre.append(step.forward_rows(model, st2.get(pending), p)[0].item())
        assert pre == plain
        n = sum(1 for _ in step.prefetch(Loader(), p, step.PinnedStager(DEV)))
        assert n == 5
    assert st.bytes_staged > 0 and all(t.is_pinned() for t in st.host[0].values())
    assert st.mode == "pinned_async"
    for mode in ("pinned", "pageable"):                                      # the other two transports give the same numbers
        stm = step.PinnedStager(DEV, depth=2, mode=mode)
        with torch.no_grad():
            got = [step.forward(model, bt, p, generator=torch.Generator().manual_seed(50 + i), stager=stm)[0].item()
                   for i, bt in enumerate(batches)]
        assert got == plain, mode
    assert torch.get_num_threads() >= 1
    assert len(set(plain)) == 5                    # the batches really differ


@pytest.mark.isolated
def test_graph_replay_follows_the_learning_rate_schedule():
    """host-side schedule work must stay OUTSIDE the captured function.  The documented pattern -- replay the
    captured device step, then `scheduler_step()` + `upload_lr()` on the host -- applies a different learning rate every step
    (warm-up ramp, then decay) exactly like the eager loop; capturing the scheduler call inside the step would freeze it."""
    from gst_visdial_amd.optim import FusedAdamW
    from gst_visdial_amd.graph import GraphedStep
    s = sc()
    g = load_npz("tiny_train.npz")

    def run(graphed, n=7):
        model, params, cfg = s.build_tiny_model("fp32", DEV, seed=9)
        model.eval()
        kw = s.golden_batch(g, DEV)
        opt = FusedAdamW(model, lr=5e-3, warmup_steps=3, t_total=12, min_lr=1e-5)
        lrs = []

        def device_step():
            loss, _ = model(**kw)
            loss.backward()
            opt.step()
            opt.zero_grad()
            return loss

        def host_end():
            lrs.append(opt.current_lrs()[0])
            opt.scheduler_step()
            opt.upload_lr()

        if graphed:
            for _ in range(2):
                device_step(); host_end()
            step = GraphedStep(device_step, warmup=0)
            for _ in range(n - 2):
                step(); host_end()
        else:
            for _ in range(n):
                device_step(); host_end()
        torch.cuda.synchronize()
        return lrs, model.engine.flat.P.clone()

    lr_e, p_e = run(False)
    lr_g, p_g = run(True)
    assert lr_e == lr_g and len(set(lr_e)) >= 5                      # the schedule really moves from step to step
    assert maxerr(p_e, p_g) < 1e-6
