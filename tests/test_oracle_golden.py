"""The CPU oracle (oracle/vd_oracle.py) against the golden vectors that
oracle/make_golden.py produced by RUNNING THE REFERENCE.  fp32 tolerance 2e-5 abs on
activations/logits (same arithmetic, different op order), id/index work bit-exact."""
import json
import os

import torch

from conftest import GOLDEN, batch_from_golden
from oracle import vd_oracle as O

TOL = 2e-5


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    assert err <= tol, err


def test_train_branch_stages(tiny_cfg, tiny_state, tiny_train):
    enc, dec = tiny_cfg
    out = O.model_forward(tiny_state, enc, dec, batch_from_golden(tiny_train))
    for k in ("enc_hidden_t", "enc_hidden_v", "enc_hidden", "dec_hidden", "logits"):
        close(out[k], tiny_train[k])
    assert torch.equal(out["enc_mask"], tiny_train["enc_mask"])
    close(out["loss"], tiny_train["loss"], 1e-5)


def test_loss_reduction_none(tiny_cfg, tiny_state, tiny_train):
    enc, dec = tiny_cfg
    out = O.model_forward(tiny_state, enc, dec, batch_from_golden(tiny_train), loss_reduction=False)
    close(out["loss"], tiny_train["loss_none"], 1e-5)
    # ignored (label 0) positions carry exactly zero loss
    lab = tiny_train["in::dec_labels"].reshape(-1)
    assert torch.all(out["loss"][lab == 0] == 0)


def test_gradients(tiny_cfg, tiny_state, tiny_train):
    enc, dec = tiny_cfg
    keys = [k[6:] for k in tiny_train if k.startswith("grad::")]
    _, g, dfeat = O.grads(tiny_state, enc, dec, batch_from_golden(tiny_train), keys)
    for k in keys:
        ref = tiny_train["grad::" + k]
        scale = max(ref.abs().max().item(), 1e-3)
        close(g[k] / scale, ref / scale, 5e-5)
    close(dfeat, tiny_train["d_feats"], 1e-6)


def test_dead_parameters_match_reference(tiny_state):
    with open(os.path.join(GOLDEN, "tiny_nograd_keys.json")) as f:
        nograd = set(json.load(f))
    # named_parameters() of the reference omits the decoder.* embedding aliases and lm_head.decoder.bias
    live = set(O.live_param_keys(tiny_state))
    named = {k for k in tiny_state
             if not k.startswith("decoder.decoder.bert.embeddings.") and k != "decoder.decoder.lm_head.decoder.bias"
             and k != "encoder.bert_pretrained.cls.predictions.decoder.weight"}
    assert named - live == nograd - {"encoder.bert_pretrained.cls.predictions.decoder.weight"}
    assert len(nograd) == 42 - 0 or len(nograd) > 0


def test_eval_branch_label_shift_and_scores(tiny_cfg, tiny_state, tiny_train, tiny_eval):
    enc, dec = tiny_cfg
    b = batch_from_golden(tiny_train, dec_key="in::eval_dec_input_ids", with_labels=False)
    unmutated = b["dec_input_ids"].clone()
    out = O.model_forward(tiny_state, enc, dec, b)
    assert torch.equal(b["dec_input_ids"], tiny_eval["mutated_ids"])        # in-place eos->pad
    close(out["logits"], tiny_eval["logits"])
    close(out["loss"], tiny_eval["loss"], 1e-5)
    close(O.answer_scores(out["logits"], unmutated), tiny_eval["scores"], 1e-4)


def test_sampling_decode_argmax(tiny_cfg, tiny_state, tiny_train, tiny_decode):
    enc, dec = tiny_cfg
    b = batch_from_golden(tiny_train)
    b["dec_input_ids"] = torch.full((b["enc_input_ids"].shape[0], 1), 101, dtype=torch.long)
    seq, trace = O.sampling_decode(tiny_state, enc, dec, b, temperature=0.7, top_k=7, top_p=0.0,
                                   ngram_blocking_size=2, draw=lambda p: p.argmax(-1, keepdim=True))
    assert torch.equal(seq, tiny_decode["sequence"])
    tr = torch.stack(trace, 0)
    ref = tiny_decode["step_logits"]
    # rows are compared up to (and including) the step that emitted [SEP]; later steps are
    # discarded by the reference (visual_dialog_model.py:113-119) and may follow a different
    # argmax trajectory on fp32 near-ties
    live = torch.ones(tr.shape[:2], dtype=torch.bool)
    for b_ in range(seq.shape[0]):
        hit = (seq[b_] == 102).nonzero()
        if hit.numel():
            live[hit[0, 0] + 1:, b_] = False
    assert live.sum() > live.numel() // 3
    kept = ref > -1e29
    assert torch.equal((~torch.isinf(tr))[live], kept[live])                 # same filtered set, bit-exact
    sel = kept & live[..., None]
    close(tr[sel], ref[sel], 1e-4)


def test_decoding_filters(utils_golden):
    u = utils_golden

    def fin(t):
        return torch.where(torch.isinf(t), torch.full_like(t, -1e30), t)

    assert torch.equal(fin(O.top_k_top_p_filter(u["logits"].clone(), top_k=5)), u["topk5"])
    assert torch.equal(fin(O.top_k_top_p_filter(u["logits"].clone(), top_k=0, top_p=0.6)), u["topp06"])
    assert torch.equal(fin(O.ngram_block(u["logits"].clone(), u["hist"], u["dec"], 3)), u["ngram3"])
    assert torch.equal(fin(O.ngram_block(u["logits"].clone(), u["hist"], u["dec"], 2)), u["ngram2"])


def test_metrics_and_schedule(utils_golden):
    u = utils_golden
    assert torch.equal(O.scores_to_ranks(u["scores"]), u["ranks"])
    m = O.sparse_metrics(O.gt_ranks(u["scores"], u["gt"]))
    got = torch.tensor([m[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")], dtype=torch.float64)
    close(got, u["sparse"].double(), 1e-6)
    nd = O.ndcg_batch(u["scores"][:, 0], u["relevance"]).mean()
    close(nd.reshape(1).double(), u["ndcg"].double(), 1e-6)
    lrs = [O.warmup_linear_nonzero_lr(s, 2e-5, 10, 40) for s in range(45)]
    close(torch.tensor(lrs, dtype=torch.float64), u["lrs"].double(), 1e-12)


def test_pad_after_eos():
    s = torch.tensor([[5, 102, 7, 102], [1, 2, 3, 4], [102, 9, 9, 9]])
    assert O.pad_after_eos(s).tolist() == [[5, 102, 0, 0], [1, 2, 3, 4], [102, 0, 0, 0]]


def test_schedule_order_full_config():
    cfg = dict(v_biattention_id=[0, 1, 2, 3, 4, 5], t_biattention_id=[6, 7, 8, 9, 10, 11],
               v_num_hidden_layers=6, num_hidden_layers=12)
    order = O.encoder_schedule(cfg)
    assert order[:7] == [("t", i) for i in range(6)] + [("c", 0)]
    assert order[7:10] == [("v", 0), ("t", 6), ("c", 1)]
    assert order[-2:] == [("v", 5), ("t", 11)]
    assert len(order) == 24


# ---- round-2 fixtures (oracle/make_golden_r2.py: the reference run on an eval set / under real sampling / in train_gen's loop)
def _expanded_eval_rows(ev, tag=""):
    """evaluate_gen.py:45-92: [B, rounds, options, L] -> one row per (dialog, round, option), image tensors expanded."""
    ids = ev[tag + "in::enc_input_ids"]
    B, NR, G, T = ids.shape
    n = B * NR * G
    dial = torch.arange(B).repeat_interleave(NR * G)
    b = dict(enc_input_ids=ids.reshape(n, T), enc_segments=ev[tag + "in::enc_segments"].reshape(n, T),
             enc_attention_mask=ev[tag + "in::enc_att_mask"].reshape(n, T),
             enc_image_features=ev[tag + "in::enc_image_feat"][dial], enc_image_spatials=ev[tag + "in::enc_image_loc"][dial],
             enc_image_mask=ev[tag + "in::enc_image_mask"][dial],
             dec_input_ids=ev[tag + "in::dec_input_ids"].reshape(n, -1).clone(),
             dec_attention_mask=ev[tag + "in::dec_att_mask"].reshape(n, -1), dec_labels=None)
    return b, (B, NR, G)


def test_eval_set_scores_ranks_and_metrics(tiny_cfg, tiny_state):
    from conftest import load_npz
    enc, dec = tiny_cfg
    ev = load_npz("tiny_evalset.npz")
    for tag in ("", "attacked::"):
        b, (B, NR, G) = _expanded_eval_rows(ev, tag)
        unmutated = b["dec_input_ids"].clone()
        out = O.model_forward(tiny_state, enc, dec, b)
        scores = O.answer_scores(out["logits"], unmutated).view(B, NR, G)
        close(scores, ev[tag + "scores"], 1e-4)
        assert ev[tag + "min_score_gap"].item() > 1e-2
        assert torch.equal(O.scores_to_ranks(scores), ev[tag + "ranks"])                    # index work: bit-exact
        m = O.sparse_metrics(O.gt_ranks(scores, ev[tag + "in::gt_option_inds"]))
        got = torch.tensor([m[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")], dtype=torch.float64)
        close(got, ev[tag + "sparse"], 1e-6)
        rid = ev[tag + "in::round_id"].squeeze(1)
        nd = O.ndcg_batch(scores[torch.arange(B), rid - 1, :], ev[tag + "in::gt_relevance"]).mean()
        close(nd.double().reshape(1), ev[tag + "ndcg"].double(), 1e-6)
    assert not torch.equal(ev["scores"], ev["attacked::scores"])


def test_sampling_decode_under_recorded_uniforms(tiny_cfg, tiny_state, tiny_train):
    """Real sampling (temperature 1.3, top-k 40): the reference drew by inverse CDF from the recorded uniforms."""
    from conftest import load_npz
    enc, dec = tiny_cfg
    sm = load_npz("tiny_sampled.npz")
    assert sm["margin"].item() > 5e-4                     # every draw sits >= 5e-4 inside its CDF cell: 1e-4 logit noise cannot flip it
    b = batch_from_golden(tiny_train)
    b["dec_input_ids"] = torch.full((b["enc_input_ids"].shape[0], 1), 101, dtype=torch.long)
    u, t = sm["uniforms"], [0]

    def draw(prob):
        c = torch.cumsum(prob.float(), dim=-1)
        x = u[t[0]].reshape(-1, 1) * c[:, -1:]
        t[0] += 1
        return (c < x).sum(-1, keepdim=True).clamp(max=prob.shape[-1] - 1)

    seq, _ = O.sampling_decode(tiny_state, enc, dec, b, temperature=float(sm["temperature"]), top_k=int(sm["top_k"]), top_p=0.0,
                               ngram_blocking_size=2, draw=draw)
    assert torch.equal(seq, sm["sequence"])
    assert (seq != 0).sum() > 30 and len(set(seq.reshape(-1).tolist())) > 20             # a real, varied sample


def test_train_loop_with_restated_adamw_follows_the_reference_run(tiny_cfg, tiny_state, tiny_train):
    """train_gen.py:300-329 (iteration-0 quirk, per-tensor groups, two learning rates, warm-up schedule) restated on the
    oracle's autograd + oracle/ref_adamw.py reproduces the losses and the parameters of the reference model's own run."""
    from conftest import load_npz
    from oracle import ref_adamw
    enc, dec = tiny_cfg
    tr = load_npz("tiny_trainer.npz")
    meta = json.load(open(os.path.join(GOLDEN, "tiny_trainer.json")))
    sd = {k: v.clone() for k, v in tiny_state.items()}
    keys = [k for k in O.live_param_keys(sd) if k in sd]
    params = {k: torch.nn.Parameter(sd[k].clone()) for k in keys}
    lang = set(meta["language_weights"])

    def hp(k):           # the group the reference put this tensor in (names relative to the encoder / decoder module)
        rel = k.split(".", 1)[1]
        if k == "decoder.decoder.lm_head.decoder.weight":
            rel = "decoder.bert.embeddings.word_embeddings.weight"
        lr = meta["lr"] if rel in lang else meta["image_lr"]
        return lr, (0.0 if any(nd in rel for nd in ref_adamw.NO_DECAY) else 0.01)

    # the reference hands ONLY the encoder's and the decoder's tensors to its optimizer (train_gen.py:209-245): the VLFusion
    # projections live on EncoderDecoderModel itself and are never updated
    assert not any("vlfusion" in n for n in meta["param_names"])
    groups = [dict(params=[params[k]], lr=hp(k)[0], weight_decay=hp(k)[1]) for k in keys if not k.startswith("vlfusion.")]
    opt = ref_adamw.AdamW(groups, lr=meta["lr"])
    base = [g["lr"] for g in groups]
    cpu_b = batch_from_golden(tiny_train)
    losses = []
    for it in range(6):
        cur = dict(sd)
        cur.update({k: p.data for k, p in params.items()})
        out, grads, _ = O.grads(cur, enc, dec, cpu_b, keys, wrt_feats=False)
        for k in keys:
            if grads[k] is not None:
                params[k].grad = grads[k] if params[k].grad is None else params[k].grad + grads[k]
        for g, b0 in zip(opt.param_groups, base):
            g["lr"] = O.warmup_linear_nonzero_lr(it, b0, meta["warmup_steps"], meta["t_total"])
        if it > 0:
            opt.step()
            opt.zero_grad()
        losses.append(out["loss"].item())
        if it == 3:
            osd = opt.state_dict()
            i198 = meta["param_names"].index("decoder.decoder.bert.embeddings.word_embeddings.weight")
            j = [k for k in keys if not k.startswith("vlfusion.")].index("decoder.decoder.lm_head.decoder.weight")
            ref_m = tr["opt3::%d::exp_avg" % i198]
            close(osd["state"][j]["exp_avg"] / ref_m.abs().max(), ref_m / ref_m.abs().max(), 2e-3)
            assert int(tr["opt3::%d::step" % i198]) == osd["state"][j]["step"] == 3
    close(torch.tensor(losses, dtype=torch.float64), tr["losses"].double(), 2e-5)
    assert losses[0] == losses[1]                                   # nothing moved at iteration 0
    worst = max((params[k].data - tr["state5::" + k]).abs().max().item() / max(tr["state5::" + k].abs().max().item(), 1e-6) for k in keys)
    assert worst < 2e-4, worst
    assert torch.equal(tr["state5::vlfusion.fc_v.weight"], tiny_state["vlfusion.fc_v.weight"])       # never trained by the reference


# ---- round 3: TRAIN mode (dropout on) under injected masks, oracle/make_golden_r3.py -------------------------------------
def _dropout_fixture():
    from conftest import load_npz
    g = load_npz("tiny_train_dropout.npz")
    with open(os.path.join(GOLDEN, "tiny_cfg_dropout.json")) as f:
        cfg = json.load(f)
    return g, cfg


def test_train_mode_under_injected_masks_matches_the_reference(tiny_state, tiny_train):
    """The reference ran model.train() with every nn.Dropout / functional dropout call replaced by x * mask / (1 - p_site)
    (p_site = what the reference's own call site passed).  The oracle, given the same masks by site label, must reproduce
    loss / logits / fused encoder states / the 29 golden-key gradients -- with ITS OWN probability per site, read from the
    config the way the reference's modules were constructed.  The config gives every dropout family a different value, so
    this pins which probability belongs to which site (incl. the decoder's embedding dropout = the ENCODER's value)."""
    g, cfg = _dropout_fixture()
    masks = O.DropMasks({k[6:]: v for k, v in g.items() if k.startswith("mask::")})
    keys = [k[6:] for k in g if k.startswith("grad::")]
    out, gr, dfeat = O.grads(tiny_state, cfg["enc"], cfg["dec"], batch_from_golden(tiny_train), keys, train=masks)
    assert masks.used == cfg["site_order"]                      # every site of the reference's step, in its order, once
    for lab in masks.used:
        assert abs(masks.p_used[lab] - g["p::" + lab].item()) < 1e-12, (lab, masks.p_used[lab], g["p::" + lab].item())
    close(out["logits"], g["logits"])
    close(out["enc_hidden"], g["enc_hidden"])
    close(out["loss"], g["loss"], 1e-5)
    for k in keys:
        ref = g["grad::" + k]
        scale = max(ref.abs().max().item(), 1e-3)
        close(gr[k] / scale, ref / scale, 5e-5)
    close(dfeat, g["d_feats"], 1e-6)


def test_injected_masks_are_not_a_no_op(tiny_state, tiny_train):
    """Sanity of the hook itself: other masks -> other outputs; a missing site raises."""
    g, cfg = _dropout_fixture()
    table = {k[6:]: v for k, v in g.items() if k.startswith("mask::")}
    flipped = dict(table)
    flipped["d1.ln3"] = 1 - table["d1.ln3"]
    out = O.model_forward(tiny_state, cfg["enc"], cfg["dec"], batch_from_golden(tiny_train), train=O.DropMasks(flipped))
    assert (out["logits"] - g["logits"]).abs().max().item() > 1e-3
    del flipped["c0.attn2"]
    import pytest
    with pytest.raises(KeyError):
        O.model_forward(tiny_state, cfg["enc"], cfg["dec"], batch_from_golden(tiny_train), train=O.DropMasks(flipped))


# ---- round-4 fixture (oracle/make_golden_r4.py: a TRAINED tiny reference checkpoint scored on 8 x 10 x 100 candidates)
def test_eval_set_100_options_trained_checkpoint(tiny_cfg):
    """The oracle on the real-shaped evaluation set (trained tiny checkpoint, 100 options per round, dense relevance): scores of the
    first two dialogs (2000 rows) to 1e-4, their ranks bit-exact; the metric restatements on the reference's full score tensor."""
    from conftest import load_npz
    from gst_visdial_amd.selfcheck import evalset100_batches
    enc, dec = tiny_cfg
    ev = load_npz("tiny_evalset100.npz")
    sd = load_npz("tiny_state_trained.npz")
    assert ev["min_score_gap"].item() >= 2e-3 - 1e-6            # ranks are a statement about the model, not about summation order
    batch = evalset100_batches(ev, dialogs_per_batch=2)[0]
    B, NR, G, T = batch["enc_input_ids"].shape
    n = B * NR * G
    dial = torch.arange(B).repeat_interleave(NR * G)
    b = dict(enc_input_ids=batch["enc_input_ids"].reshape(n, T), enc_segments=batch["enc_segments"].reshape(n, T),
             enc_attention_mask=batch["enc_att_mask"].reshape(n, T), enc_image_features=batch["enc_image_feat"][dial],
             enc_image_spatials=batch["enc_image_loc"][dial], enc_image_mask=batch["enc_image_mask"][dial],
             dec_input_ids=batch["dec_input_ids"].reshape(n, -1).clone(), dec_attention_mask=batch["dec_att_mask"].reshape(n, -1),
             dec_labels=None)
    unmutated = b["dec_input_ids"].clone()
    with torch.no_grad():
        out = O.model_forward(sd, enc, dec, b)
    scores = O.answer_scores(out["logits"], unmutated).view(B, NR, G)
    close(scores, ev["scores"][:B], 1e-4)
    assert torch.equal(O.scores_to_ranks(scores), ev["ranks"][:B].long())
    full = ev["scores"]
    assert torch.equal(O.scores_to_ranks(full), ev["ranks"].long())
    m = O.sparse_metrics(O.gt_ranks(full, ev["in::gt_option_inds"]))
    close(torch.tensor([m[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")], dtype=torch.float64), ev["sparse"], 1e-6)
    rid = ev["in::round_id"].squeeze(1)
    nd = O.ndcg_batch(full[torch.arange(full.shape[0]), rid - 1, :], ev["in::gt_relevance"]).mean()
    close(nd.double().reshape(1), ev["ndcg"].double(), 1e-6)
    # a checkpoint worth the name: far from the 1 % / 5 % / 10 % of a random ranking
    assert ev["sparse"][1].item() > 0.3 and ev["sparse"][2].item() > 0.5
