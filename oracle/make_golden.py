"""Generate tests/golden/*.npz by importing and running the REAL reference on CPU.

Run only in the build container:  python oracle/make_golden.py
(the reference's Python never travels; only the arrays written here do).

Fixtures written (all fp32 / int64 numpy arrays):
  tiny_state.npz     seeded state_dict of the reference EncoderDecoderModel (tiny config)
  tiny_cfg.json      {"enc": {...}, "dec": {...}} the two model config dicts
  tiny_train.npz     inputs + eval()-mode outputs of the train-labels branch
                     (models/visual_dialog_model.py:24-72): enc_hidden_t/v, fused states+mask,
                     decoder hidden, logits, loss(mean), loss(none), selected grads,
                     d loss / d enc_image_features
  tiny_eval.npz      labels=None branch (models/visual_dialog_decoder.py:53-57): mutated ids,
                     loss, logits, per-candidate scores (evaluate_gen.py:94-106)
  tiny_decode.npz    18-step sampling decode with torch.multinomial replaced by argmax:
                     per-step filtered logits and the final padded sequence
                     (models/visual_dialog_model.py:74-120)
  utils.npz          batch_top_k_top_p_sampling / batch_ngram_blocking on fixed logits,
                     scores_to_ranks / SparseGTMetrics / NDCG on fixed scores,
                     WarmupLinearScheduleNonZero lr sequence
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

GRAD_KEYS = [
    "encoder.bert_pretrained.bert.embeddings.word_embeddings.weight",
    "encoder.bert_pretrained.bert.embeddings.position_embeddings.weight",
    "encoder.bert_pretrained.bert.embeddings.token_type_embeddings.weight",
    "encoder.bert_pretrained.bert.embeddings.token_type_embeddings_extension.weight",
    "encoder.bert_pretrained.bert.embeddings.LayerNorm.weight",
    "encoder.bert_pretrained.bert.v_embeddings.image_embeddings.weight",
    "encoder.bert_pretrained.bert.v_embeddings.image_location_embeddings.weight",
    "encoder.bert_pretrained.bert.v_embeddings.image_location_embeddings.bias",
    "encoder.bert_pretrained.bert.encoder.layer.0.attention.self.query.weight",
    "encoder.bert_pretrained.bert.encoder.layer.0.attention.self.key.bias",
    "encoder.bert_pretrained.bert.encoder.layer.3.output.dense.bias",
    "encoder.bert_pretrained.bert.encoder.layer.2.intermediate.dense.weight",
    "encoder.bert_pretrained.bert.encoder.v_layer.1.attention.self.key.weight",
    "encoder.bert_pretrained.bert.encoder.v_layer.0.output.LayerNorm.bias",
    "encoder.bert_pretrained.bert.encoder.c_layer.0.biattention.query2.weight",
    "encoder.bert_pretrained.bert.encoder.c_layer.0.biattention.value1.weight",
    "encoder.bert_pretrained.bert.encoder.c_layer.1.biOutput.dense1.weight",
    "encoder.bert_pretrained.bert.encoder.c_layer.1.biOutput.dense2.bias",
    "encoder.bert_pretrained.bert.encoder.c_layer.0.t_output.LayerNorm.weight",
    "encoder.bert_pretrained.bert.encoder.c_layer.1.v_intermediate.dense.weight",
    "vlfusion.fc_v.weight",
    "vlfusion.fc_l.bias",
    "decoder.decoder.bert.encoder.layer.0.attention.self.value.weight",
    "decoder.decoder.bert.encoder.layer.0.crossattention.self.key.weight",
    "decoder.decoder.bert.encoder.layer.1.crossattention.self.query.bias",
    "decoder.decoder.bert.encoder.layer.1.intermediate.dense.weight",
    "decoder.decoder.bert.encoder.layer.1.output.LayerNorm.weight",
    "decoder.decoder.lm_head.decoder.weight",
    "decoder.decoder.lm_head.bias",
]


def make_inputs(seed=7, B=3, T=24, R=7, U=9, V=320, F=40):
    g = torch.Generator().manual_seed(seed)
    lens = [T, 17, 11][:B]
    ids = torch.zeros(B, T, dtype=torch.long)
    seg = torch.zeros(B, T, dtype=torch.long)
    for b, L in enumerate(lens):
        ids[b, :L] = torch.randint(104, V, (L,), generator=g)
        ids[b, 0] = 101
        cur = 1
        for p in range(3, L, 4):       # [SEP] at the end of every "utterance"
            ids[b, p] = 102
        for t in range(L):
            seg[b, t] = cur
            if ids[b, t] == 102:
                cur = 1 - cur
    seg[1, 3:6] = 2                     # exercise token_type_embeddings_extension (vilbert_dialog.py:334-347)
    seg[1, 6] = 3
    att = (ids != 0).float()
    feats = torch.randn(B, R, F, generator=g).abs()
    feats[:, 0] = feats[:, 1:].mean(1)
    loc = torch.rand(B, R, 5, generator=g)
    loc[:, 0] = torch.tensor([0., 0., 1., 1., 1.])
    img_mask = torch.ones(B, R)
    img_mask[2, -2:] = 0                # an image-padded row
    feats[2, -2:] = 0
    loc[2, -2:] = 0
    alens = [5, 3, 7][:B]
    dec_ids = torch.zeros(B, U, dtype=torch.long)
    dec_labels = torch.zeros(B, U, dtype=torch.long)
    dec_att = torch.zeros(B, U)
    eval_dec_ids = torch.zeros(B, U, dtype=torch.long)
    for b, L in enumerate(alens):
        ans = torch.randint(104, V, (L,), generator=g)
        dec_ids[b, 0] = 101
        dec_ids[b, 1:1 + L] = ans
        dec_labels[b, :L] = ans
        dec_labels[b, L] = 102
        dec_att[b, :L + 2] = 1
        eval_dec_ids[b, 0] = 101
        eval_dec_ids[b, 1:1 + L] = ans
        eval_dec_ids[b, 1 + L] = 102
    return dict(enc_input_ids=ids, enc_segments=seg, enc_attention_mask=att,
                enc_image_features=feats, enc_image_spatials=loc, enc_image_mask=img_mask,
                dec_input_ids=dec_ids, dec_attention_mask=dec_att, dec_labels=dec_labels,
                eval_dec_input_ids=eval_dec_ids)


def call_model(model, inp, dec_ids, dec_labels, loss_reduction=True, **kw):
    B, T = inp["enc_input_ids"].shape
    return model(
        enc_image_features=inp["enc_image_features"], enc_image_spatials=inp["enc_image_spatials"],
        enc_image_mask=inp["enc_image_mask"], enc_image_target=None, enc_image_label=None,
        enc_next_sentence_labels=None, enc_input_ids=inp["enc_input_ids"], enc_segments=inp["enc_segments"],
        enc_sep_indices=torch.zeros(B, 5, dtype=torch.long), enc_mlm_labels=torch.full((B, T), -1),
        enc_attention_mask=inp["enc_attention_mask"], dec_input_ids=dec_ids,
        dec_attention_mask=inp["dec_attention_mask"], dec_labels=dec_labels, loss_reduction=loss_reduction, **kw)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    cfg_dir = os.path.join(OUT, "_cfg")
    e_path, d_path = rh.write_tiny_configs(cfg_dir)
    model, params = rh.build_reference_model(e_path, d_path, mode="vd_train", seed=0)
    model.eval()
    # make LayerNorm / bias parameters non-trivial so parity tests see them
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("LayerNorm.weight") or "LayerNorm1.weight" in n or "LayerNorm2.weight" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias") or n.endswith("lm_head.bias"):
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    sd = model.state_dict()
    np.savez(os.path.join(OUT, "tiny_state.npz"), **npy(sd))
    with open(os.path.join(OUT, "tiny_cfg.json"), "w") as f:
        json.dump({"enc": rh.TINY_ENC_CFG, "dec": rh.TINY_DEC_CFG}, f, indent=1, sort_keys=True)

    inp = make_inputs()

    # ---- train-labels branch, eval() so dropout is off ---------------------------------
    captured = {}
    h1 = model.encoder.register_forward_hook(lambda m, i, o: captured.update(enc_t=o[5], enc_v=o[6]))
    h2 = model.vlfusion.register_forward_hook(lambda m, i, o: captured.update(fused=o[0], fused_mask=o[1]))
    h3 = model.decoder.decoder.bert.register_forward_hook(lambda m, i, o: captured.update(dec_hidden=o[0]))
    feats = inp["enc_image_features"].clone().requires_grad_(True)
    inp_g = dict(inp, enc_image_features=feats)
    model.zero_grad()
    loss, logits = call_model(model, inp_g, inp["dec_input_ids"].clone(), inp["dec_labels"])
    loss.backward()
    named = dict(model.named_parameters())
    remap = {"decoder.decoder.lm_head.bias": "decoder.decoder.lm_head.bias"}
    grads = {}
    for k in GRAD_KEYS:
        p = named.get(k)
        if p is None:   # decoder.* alias names are de-duplicated by named_parameters
            raise KeyError(k)
        grads["grad::" + k] = p.grad.clone()
    nograd = sorted(n for n, p in named.items() if p.grad is None)
    with torch.no_grad():
        loss_none, _ = call_model(model, inp, inp["dec_input_ids"].clone(), inp["dec_labels"], loss_reduction=False)
    out = dict(loss=loss, logits=logits, loss_none=loss_none, d_feats=feats.grad,
               enc_hidden_t=captured["enc_t"], enc_hidden_v=captured["enc_v"], enc_hidden=captured["fused"],
               enc_mask=captured["fused_mask"], dec_hidden=captured["dec_hidden"])
    out.update(grads)
    out.update({"in::" + k: v for k, v in inp.items()})
    np.savez(os.path.join(OUT, "tiny_train.npz"), **npy(out))
    with open(os.path.join(OUT, "tiny_nograd_keys.json"), "w") as f:
        json.dump(nograd, f, indent=0)

    # ---- labels=None branch (eval scoring) ---------------------------------------------
    params["mode"] = "vd_eval_val"
    with torch.no_grad():
        ids_mut = inp["eval_dec_input_ids"].clone()
        loss_e, logits_e = call_model(model, inp, ids_mut, None)
        import torch.nn.functional as F
        lp = F.log_softmax(logits_e, dim=-1)
        tgt = inp["eval_dec_input_ids"].new_zeros(inp["eval_dec_input_ids"].shape)
        tgt[:, :-1] = inp["eval_dec_input_ids"][:, 1:].clone()
        s = torch.gather(lp, -1, tgt.unsqueeze(-1)).squeeze(-1)
        scores = (s * (tgt != 0).float()).sum(-1)           # evaluate_gen.py:94-106 run verbatim here
    np.savez(os.path.join(OUT, "tiny_eval.npz"), **npy(dict(loss=loss_e, logits=logits_e, mutated_ids=ids_mut,
                                                            scores=scores)))

    # ---- sampling decode with multinomial -> argmax ------------------------------------
    params["mode"] = "vd_gen_val"
    import models.visual_dialog_model as M
    trace = []
    orig_filter = M.batch_top_k_top_p_sampling
    orig_multi = torch.multinomial

    def rec_filter(logits, **kw):
        o = orig_filter(logits, **kw)
        trace.append(o.clone())
        return o

    M.batch_top_k_top_p_sampling = rec_filter
    torch.multinomial = lambda prob, n, **kw: prob.argmax(-1, keepdim=True)
    try:
        with torch.no_grad():
            start = torch.full((inp["enc_input_ids"].shape[0], 1), 101, dtype=torch.long)
            seq = call_model(model, inp, start, None, temperature=0.7, top_k=7, top_p=0.0, ngram_blocking_size=2)
    finally:
        torch.multinomial = orig_multi
        M.batch_top_k_top_p_sampling = orig_filter
    tr = torch.stack(trace, 0)
    tr = torch.where(torch.isinf(tr), torch.full_like(tr, -1e30), tr)
    np.savez(os.path.join(OUT, "tiny_decode.npz"), **npy(dict(sequence=seq, step_logits=tr)))
    params["mode"] = "vd_train"

    # ---- utilities -----------------------------------------------------------------------
    du, vm, ou = rh.reference_utils()
    g = torch.Generator().manual_seed(3)
    lg = torch.randn(4, 50, generator=g)
    topk = du.batch_top_k_top_p_sampling(lg.clone(), top_k=5, top_p=0.0)
    topp = du.batch_top_k_top_p_sampling(lg.clone(), top_k=0, top_p=0.6)
    hist = torch.tensor([[101, 7, 8, 9, 102, 7, 8, 11, 0, 0], [101, 5, 6, 5, 6, 7, 102, 0, 0, 0],
                         [0] * 10, [101, 20, 21, 22, 23, 24, 25, 26, 27, 102]])
    dec = torch.tensor([[101, 7, 8], [101, 5, 6], [101, 3, 4], [101, 22, 23]])
    ng3 = du.batch_ngram_blocking(lg.clone(), hist, dec, ngram_size=3)
    ng2 = du.batch_ngram_blocking(lg.clone(), hist, dec, ngram_size=2)
    sc = torch.randn(3, 2, 10, generator=g)
    sc[0, 0, 3] = sc[0, 0, 5]          # a tie
    ranks = vm.scores_to_ranks(sc.clone())
    gt = torch.randint(0, 10, (3, 2), generator=g)
    sp = vm.SparseGTMetrics()
    sp.observe(sc.clone(), gt)
    spm = sp.retrieve(reset=True)
    rel = (torch.rand(3, 10, generator=g) > 0.6).float() * torch.rand(3, 10, generator=g)
    rel[:, 0] = 1.0
    nd = vm.NDCG()
    nd.observe(sc[:, 0].clone(), rel)
    ndm = nd.retrieve(reset=True)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=2e-5)
    sch = ou.WarmupLinearScheduleNonZero(opt, warmup_steps=10, t_total=40)
    lrs = []
    for _ in range(45):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()

    def fin(t):
        return torch.where(torch.isinf(t), torch.full_like(t, -1e30), t)

    np.savez(os.path.join(OUT, "utils.npz"), **npy(dict(
        logits=lg, topk5=fin(topk), topp06=fin(topp), hist=hist, dec=dec, ngram3=fin(ng3), ngram2=fin(ng2),
        scores=sc, ranks=ranks, gt=gt, sparse=np.array([spm[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")]),
        relevance=rel, ndcg=np.array([ndm["ndcg"]]), lrs=np.array(lrs))))
    import shutil
    shutil.rmtree(cfg_dir)
    print("golden written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print("  %-28s %8d bytes" % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == "__main__":
    main()
