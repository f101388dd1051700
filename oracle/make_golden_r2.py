"""Round-2 golden vectors, again produced by importing and RUNNING the real reference on CPU (build container only):

    python oracle/make_golden_r2.py

Adds to tests/golden/ (the round-1 fixtures of oracle/make_golden.py are left untouched; same seeded tiny model):

  tiny_evalset.npz   evaluate_gen.py:45-118 on 2 dialogs x 3 rounds x 5 answer options: the eval dataloader's layout
                     ([B, rounds, options, L] text tensors, per-dialog image tensors), every [dialog, round, option] row
                     through the reference model the reference's way (expanded image tensors, labels=None branch), the
                     scoring arithmetic of evaluate_gen.py:94-106 on its logits, then the reference's own
                     scores_to_ranks / SparseGTMetrics / NDCG.  A second copy has ONE option's context perturbed (what
                     attack='random_token' does per option row) -> contexts differ inside a round.
  tiny_sampled.npz   the 18-step sampling decode (models/visual_dialog_model.py:74-120) under REAL sampling: the
                     reference's torch.multinomial is replaced by an inverse-CDF draw from recorded uniforms (multinomial's
                     own stream is device specific), temperature 1.3 / top-k 40; sequence, uniforms, the smallest distance
                     of any draw to a CDF edge (the test asserts it is far above the 1e-4 logit tolerance).
  tiny_trainer.npz   train_gen.py's loop (:300-329) on the reference model with the restated pytorch_transformers AdamW
  + tiny_trainer.json (oracle/ref_adamw.py) and the reference's WarmupLinearScheduleNonZero: 6 iterations incl. the
                     iteration-0 quirk (backward only: no step, no zero_grad, so iteration 1 steps on the SUM of two
                     gradients), per-tensor param groups built BEFORE the embedding aliasing, two learning rates;
                     losses, model state after iterations 3 and 5, the optimizer's state_dict() after iteration 3 in the
                     reference's on-disk layout (what '-continue' loads, train_gen.py:254-276).  eval() mode: dropout off
                     (the reference trains with dropout; its RNG stream cannot be shared with a device).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh            # noqa: E402
from oracle import ref_adamw                    # noqa: E402
from oracle.make_golden import make_inputs, call_model, npy   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_state():
    with np.load(os.path.join(OUT, "tiny_state.npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def eval_batch(seed=21, B=2, NR=3, G=5, T=24, R=7, U=9, V=320, Fdim=40):
    """A batch as the eval dataloader emits it (dataloader_visdial_gen.py:379-388)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(B, NR, G, T, dtype=torch.long)
    seg = torch.zeros(B, NR, G, T, dtype=torch.long)
    dec = torch.zeros(B, NR, G, U, dtype=torch.long)
    datt = torch.zeros(B, NR, G, U)
    for b in range(B):
        L = 6 + 2 * b
        ctx = torch.randint(104, V, (T,), generator=g)
        for r in range(NR):
            L = min(T, L + 4)                       # the history grows round by round
            row = ctx.clone()
            row[0] = 101
            row[L:] = 0
            for p in range(3, L, 4):
                row[p] = 102
            cur, s = 1, torch.zeros(T, dtype=torch.long)
            for t in range(L):
                s[t] = cur
                if row[t] == 102:
                    cur = 1 - cur
            ids[b, r, :] = row
            seg[b, r, :] = s
            for o in range(G):
                n = int(torch.randint(1, U - 1, (1,), generator=g))
                ans = torch.randint(104, V, (n,), generator=g)
                dec[b, r, o, 0] = 101
                dec[b, r, o, 1:1 + n] = ans
                dec[b, r, o, 1 + n] = 102
                datt[b, r, o, :n + 2] = 1
    feats = torch.randn(B, R, Fdim, generator=g).abs()
    feats[:, 0] = feats[:, 1:].mean(1)
    loc = torch.rand(B, R, 5, generator=g)
    loc[:, 0] = torch.tensor([0., 0., 1., 1., 1.])
    imask = torch.ones(B, R)
    imask[1, -1] = 0
    feats[1, -1] = 0
    loc[1, -1] = 0
    gt = torch.randint(0, G, (B, NR), generator=g)
    rel = (torch.rand(B, G, generator=g) > 0.5).float() * torch.rand(B, G, generator=g)
    rel[:, 0] = 1.0
    round_id = torch.tensor([[2], [3]])[:B]
    return dict(enc_input_ids=ids, enc_segments=seg, enc_att_mask=(ids != 0).float(), dec_input_ids=dec, dec_att_mask=datt,
                enc_image_feat=feats, enc_image_loc=loc, enc_image_mask=imask, gt_option_inds=gt, gt_relevance=rel,
                round_id=round_id)


def reference_eval_scores(model, params, batch):
    """evaluate_gen.py:45-106 on one batch (chunking dropped: one chunk holds every row)."""
    ids = batch["enc_input_ids"]
    B, NR, G, T = ids.shape
    n = B * NR * G
    R = batch["enc_image_feat"].shape[-2]
    feats = batch["enc_image_feat"].unsqueeze(1).unsqueeze(1).expand(B, NR, G, R, -1).contiguous().view(n, R, -1)
    loc = batch["enc_image_loc"].unsqueeze(1).unsqueeze(1).expand(B, NR, G, R, 5).contiguous().view(n, R, 5)
    imask = batch["enc_image_mask"].unsqueeze(1).unsqueeze(1).expand(B, NR, G, R).contiguous().view(n, R)
    dec_ids = batch["dec_input_ids"].view(n, -1)
    inp = dict(enc_input_ids=ids.view(n, T), enc_segments=batch["enc_segments"].view(n, T),
               enc_attention_mask=batch["enc_att_mask"].view(n, T), enc_image_features=feats, enc_image_spatials=loc,
               enc_image_mask=imask, dec_attention_mask=batch["dec_att_mask"].view(n, -1))
    params["mode"] = "vd_eval_val"
    with torch.no_grad():
        _, lm_scores = call_model(model, inp, dec_ids.clone(), None)     # train_gen.forward feeds a device COPY of the ids
        lm_scores = F.log_softmax(lm_scores, dim=-1)
        tgt = dec_ids.new_zeros(dec_ids.shape)
        tgt[:, :-1] = dec_ids[:, 1:].clone()
        s = torch.gather(lm_scores, -1, tgt.unsqueeze(-1)).squeeze(-1)
        s = (s * (tgt != 0).float()).sum(-1)
    params["mode"] = "vd_train"
    return s.view(B, NR, G)


def main_evalset(model, params):
    du, vm, ou = rh.reference_utils()
    out = {}
    for tag, perturb in (("", False), ("attacked::", True)):
        batch = eval_batch()
        if perturb:                                # one option row of round 1 sees a different context
            ids = batch["enc_input_ids"]
            ids[0, 1, 3, 5:8] = torch.tensor([111, 222, 133])
            ids[1, 2, 0, 4] = 300
        scores = reference_eval_scores(model, params, batch)
        sp = vm.SparseGTMetrics()
        sp.observe(scores.clone(), batch["gt_option_inds"])
        spm = sp.retrieve(reset=True)
        nd = vm.NDCG()
        rid = batch["round_id"].squeeze(1)
        nd.observe(scores[torch.arange(scores.size(0)), rid - 1, :].clone(), batch["gt_relevance"])
        ndm = nd.retrieve(reset=True)
        ranks = vm.scores_to_ranks(scores.clone())
        srt = scores.sort(-1)[0]
        out.update({tag + "in::" + k: v for k, v in batch.items()})
        out.update({tag + "scores": scores, tag + "ranks": ranks,
                    tag + "sparse": np.array([spm[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")]),
                    tag + "ndcg": np.array([ndm["ndcg"]]),
                    tag + "min_score_gap": (srt[..., 1:] - srt[..., :-1]).min()})
    np.savez(os.path.join(OUT, "tiny_evalset.npz"), **npy(out))
    print("evalset: min gap between neighbouring candidate scores %.4f / %.4f (ranks are stable under 1e-3 noise)"
          % (out["min_score_gap"].item(), out["attacked::min_score_gap"].item()))


def main_sampled(model, params):
    inp = make_inputs()
    B = inp["enc_input_ids"].shape[0]
    steps = 18
    for seed in range(100, 400):                  # take the first uniforms whose draws all sit well inside their CDF cell
        g = torch.Generator().manual_seed(seed)
        u = torch.rand(steps, B, generator=g) * 0.98 + 0.01
        state = dict(t=0, margin=1.0)

        def draw(prob, n, **kw):
            c = torch.cumsum(prob.float(), dim=-1)
            x = u[state["t"]].reshape(-1, 1) * c[:, -1:]
            idx = (c < x).sum(-1, keepdim=True).clamp(max=prob.shape[-1] - 1)
            lo = torch.where(idx > 0, c.gather(-1, (idx - 1).clamp(min=0)), torch.zeros_like(x))
            hi = c.gather(-1, idx)
            state["margin"] = min(state["margin"], float(torch.minimum(x - lo, hi - x).min()))
            state["t"] += 1
            return idx

        params["mode"] = "vd_gen_val"
        orig = torch.multinomial
        torch.multinomial = draw
        try:
            with torch.no_grad():
                start = torch.full((B, 1), 101, dtype=torch.long)
                seq = call_model(model, inp, start, None, temperature=1.3, top_k=40, top_p=0.0, ngram_blocking_size=2)
        finally:
            torch.multinomial = orig
            params["mode"] = "vd_train"
        if state["margin"] > 5e-4:
            break
    assert state["margin"] > 5e-4, state
    np.savez(os.path.join(OUT, "tiny_sampled.npz"), **npy(dict(sequence=seq, uniforms=u, margin=np.array([state["margin"]]),
                                                               temperature=np.array([1.3]), top_k=np.array([40]))))
    print("sampled decode: seed %d, min CDF margin %.4f, sequence\n%s" % (seed, state["margin"], seq))


def main_trainer(e_path, d_path):
    mods = rh._install_shims()
    E, D = mods["E"], mods["D"]
    du, vm, ou = rh.reference_utils()
    params = dict(model_enc_config=e_path, model_dec_config=d_path, gpu_ids=[0], model="enc_dec_a", mode="vd_train",
                  batch_size=3, device=torch.device("cpu"))
    torch.manual_seed(0)
    enc, dec = E.VisualDialogEncoder(params), D.VisualDialogDecoder(params)
    model = mods["EncoderDecoderModel"](params, enc, dec)
    lr, image_lr, warm, total = 3e-3, 1e-3, 2, 50
    language = [k for k, _ in enc.named_parameters() if ".v_" not in k and "v_embeddings" not in k and "c_layer" not in k]
    groups, names = ref_adamw.reference_param_groups(enc, dec, lr, image_lr, language)     # BEFORE the aliasing, as train_gen.py does
    opt = ref_adamw.AdamW(groups, lr=lr)
    sch = ou.WarmupLinearScheduleNonZero(opt, warmup_steps=warm, t_total=total)
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings                  # train_gen.py:293
    model.load_state_dict(load_state(), strict=True)
    model.eval()
    inp = make_inputs()
    losses, lrs, states = [], [], {}
    for it in range(6):
        loss, _ = call_model(model, inp, inp["dec_input_ids"].clone(), inp["dec_labels"])
        loss.backward()
        lrs.append([opt.param_groups[0]["lr"], opt.param_groups[names.index("encoder.bert_pretrained.bert.v_embeddings.image_embeddings.weight")]["lr"]])
        if it > 0:                                   # train_gen.py:326-329
            opt.step()
            opt.zero_grad()
        sch.step()
        losses.append(loss.item())
        if it in (3, 5):
            states[it] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        if it == 3:
            osd = opt.state_dict()
            opt3 = {}
            for idx, st in osd["state"].items():
                opt3["opt3::%d::exp_avg" % idx] = st["exp_avg"].clone()
                opt3["opt3::%d::exp_avg_sq" % idx] = st["exp_avg_sq"].clone()
                opt3["opt3::%d::step" % idx] = np.array([st["step"]])
            sched3 = sch.state_dict()
            groups3 = [dict(lr=g["lr"], weight_decay=g["weight_decay"], params=g["params"],
                            initial_lr=g.get("initial_lr")) for g in osd["param_groups"]]
    out = dict(losses=np.array(losses), lrs=np.array(lrs))
    out.update({"state3::" + k: v for k, v in states[3].items()})
    out.update({"state5::" + k: v for k, v in states[5].items()})
    out.update(opt3)
    np.savez(os.path.join(OUT, "tiny_trainer.npz"), **npy(out))
    meta = dict(param_names=names, shapes=[list(g["params"][0].shape) for g in groups], groups_iter3=groups3,
                scheduler_iter3={k: v for k, v in sched3.items() if isinstance(v, (int, float, list))},
                lr=lr, image_lr=image_lr, warmup_steps=warm, t_total=total, language_weights=language,
                stateful_indices=sorted(int(i) for i in osd["state"].keys()))
    with open(os.path.join(OUT, "tiny_trainer.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("trainer: losses", ["%.5f" % x for x in losses], "| %d param groups, %d with state" % (len(names), len(osd["state"])))


def main():
    cfg_dir = os.path.join(OUT, "_cfg")
    e_path, d_path = rh.write_tiny_configs(cfg_dir)
    model, params = rh.build_reference_model(e_path, d_path, mode="vd_train", seed=0)
    model.load_state_dict(load_state(), strict=True)
    model.eval()
    main_evalset(model, params)
    main_sampled(model, params)
    main_trainer(e_path, d_path)
    import shutil
    shutil.rmtree(cfg_dir)
    for fn in sorted(os.listdir(OUT)):
        print("  %-28s %8d bytes" % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == "__main__":
    main()
