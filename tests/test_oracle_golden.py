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
