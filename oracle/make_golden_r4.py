"""Round-4 golden vectors: a REAL-SHAPED evaluation set scored by the imported reference (build container only):

    python oracle/make_golden_r4.py

  tests/golden/tiny_state_trained.npz   a TRAINED tiny checkpoint: the reference model itself (tiny config, /root/reference's
                     classes, seeded init) fitted for a few thousand Adam steps to a synthetic dialog task -- an answer's first token depends
                     on the image class (cross-attention to the vision rows), the rest follow a noisy bigram rule -- so that its answer distributions are PEAKED like a
                     trained checkpoint's, not the near-uniform ones of a random init (where all candidates of one length
                     tie to within bf16 noise and rank metrics say nothing).
  tests/golden/tiny_evalset100.npz      evaluate_gen.py:45-118 on 8 dialogs x 10 rounds x 100 answer options of HELD-OUT dialogs
                     of the same task: the eval dataloader's layout (the context stored once per round -- the 100 option rows
                     of a round carry the same context, dataloader_visdial_gen.py:379-388 -- and expanded by the test), option 0..99 =
                     the ground truth, near-duplicates of it (one token changed: the close calls that make ranks flip),
                     other rounds' answers and random answers, shuffled; dense gt_relevance for one round per dialog.
                     Every [dialog, round, option] row went through the reference model the reference's way (labels=None
                     branch, evaluate_gen.py:94-106 scoring); scores, scores_to_ranks, SparseGTMetrics and NDCG are the
                     reference's own functions' outputs.

The north_star gate this pins: "evaluate_gen.py NDCG/MRR ... within 0.1 of the reference checkpoint" -- in fp32 (ranks bit-exact)
AND in the bf16 mode that ships (metrics within 0.1 point, rank-flip rate recorded), tests/test_round4_gpu.py.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh                                  # noqa: E402
from oracle.make_golden import call_model, npy                        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
V0, V1 = 104, 320                   # ordinary tokens
T, U, R, FD = 40, 9, 7, 40
NCLS = 4                            # image classes
HIST = 3                            # rounds of history kept in the context (the tiny config has 48 positions, not 256)


def _perm(seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(V1 - V0, generator=g) + V0


PERM = _perm()


FIRST = PERM[: NCLS * 8].view(NCLS, 8)          # the 8 tokens an answer about an image of class c can start with


def make_dialog(g, noise=0.15):
    """One synthetic dialog: caption, 10 (question, answer) rounds, an image of class c.
    An answer starts with one of the 8 tokens of its image's class (readable only through the cross-attention to the vision rows)
    and continues a_{j+1} = PERM[a_j] (w.p. 1 - noise, else random): a decoder-side bigram rule.  Questions are random."""
    c = int(torch.randint(0, NCLS, (1,), generator=g))
    cap = torch.randint(V0, V1, (2,), generator=g)
    qs, ans = [], []
    for _ in range(10):
        nq = int(torch.randint(2, 4, (1,), generator=g))
        q = torch.randint(V0, V1, (nq,), generator=g)
        na = int(torch.randint(1, 5, (1,), generator=g))
        a = torch.zeros(na, dtype=torch.long)
        a[0] = FIRST[c, int(torch.randint(0, 8, (1,), generator=g))]
        for j in range(1, na):
            a[j] = PERM[a[j - 1] - V0] if float(torch.rand(1, generator=g)) >= noise else int(torch.randint(V0, V1, (1,), generator=g))
        qs.append(q)
        ans.append(a)
    feats = torch.randn(R, FD, generator=g).abs() * 0.5
    feats[1:, c * 10:(c + 1) * 10] += 1.5                      # the class is readable from the region features
    feats[0] = feats[1:].mean(0)
    loc = torch.rand(R, 5, generator=g)
    loc[0] = torch.tensor([0., 0., 1., 1., 1.])
    return dict(cls=c, cap=cap, qs=qs, ans=ans, feats=feats, loc=loc)


def context(d, r):
    """[CLS] caption [SEP] (q [SEP] a [SEP]) x last HIST rounds, q_r [SEP]; segments flip per utterance starting at 1."""
    ids, seg, cur = [101], [1], 1
    def utt(tokens):
        nonlocal cur
        for t in tokens.tolist():
            ids.append(t); seg.append(cur)
        ids.append(102); seg.append(cur)
        cur = 1 - cur
    utt(d["cap"])
    for k in range(max(0, r - HIST), r):
        utt(d["qs"][k]); utt(d["ans"][k])
    utt(d["qs"][r])
    assert len(ids) <= T, len(ids)
    row = torch.zeros(T, dtype=torch.long); s = torch.zeros(T, dtype=torch.long)
    row[:len(ids)] = torch.tensor(ids); s[:len(ids)] = torch.tensor(seg)
    return row, s


def dec_rows(a):
    """dec_input_ids = [CLS] answer with [SEP] -> 0, mask 1 through the [SEP] slot, labels = answer [SEP] (dataloader_visdial_gen.py:167-230)."""
    n = len(a)
    ids = torch.zeros(U, dtype=torch.long); att = torch.zeros(U); lab = torch.zeros(U, dtype=torch.long)
    ids[0] = 101; ids[1:1 + n] = a
    att[:n + 2] = 1
    lab[:n] = a; lab[n] = 102
    return ids, att, lab


def train_batch(g, B):
    rows = []
    for _ in range(B):
        d = make_dialog(g)
        r = int(torch.randint(0, 10, (1,), generator=g))
        ids, seg = context(d, r)
        di, da, dl = dec_rows(d["ans"][r])
        rows.append((ids, seg, d["feats"], d["loc"], di, da, dl))
    st = lambda i: torch.stack([x[i] for x in rows])
    ids = st(0)
    inp = dict(enc_input_ids=ids, enc_segments=st(1), enc_attention_mask=(ids != 0).float(), enc_image_features=st(2),
               enc_image_spatials=st(3), enc_image_mask=torch.ones(B, R), dec_attention_mask=st(5))
    return inp, st(4), st(6)


def train_checkpoint(model, params, steps=int(os.environ.get("R4_STEPS", "2000")), B=32):
    """Fit the REFERENCE model (its own forward / loss, torch autograd) to the synthetic task; dropout off (eval mode)."""
    model.eval()
    ps = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(ps, lr=1e-3)
    g = torch.Generator().manual_seed(4242)
    params["mode"] = "vd_train"
    for it in range(steps):
        inp, dec_ids, labels = train_batch(g, B)
        for gr in opt.param_groups:                       # 100-step warm-up, then two decays (below)
            gr["lr"] = gr.setdefault("base", 1e-3) * min(1.0, (it + 1) / 100.0)
        loss, _ = call_model(model, inp, dec_ids, labels)
        opt.zero_grad(set_to_none=True)
        loss.mean().backward()
        opt.step()
        if it in (steps // 2, int(steps * 0.8)):
            for gr in opt.param_groups:
                gr["base"] *= 0.3
        if it % 250 == 0 or it == steps - 1:
            print("  train step %4d  loss %.4f" % (it, loss.mean().item()), flush=True)
    return model


def eval_slots(seed=99, B=8, NR=10, G=100):
    """-> (generator, dialogs, round_id, slots[b][r] = list of (answer tokens, kind) in option order; kind 2 = ground truth,
    1 = near-duplicate of it, 0 = other)."""
    g = torch.Generator().manual_seed(seed)
    round_id = torch.randint(1, NR + 1, (B, 1), generator=g)
    dialogs = [make_dialog(g) for _ in range(B)]
    pool = [a for d in dialogs for a in d["ans"]]
    slots = []
    for b, d in enumerate(dialogs):
        per_round = []
        for r in range(NR):
            a = d["ans"][r]
            opts, kind = [a], [2]
            while len(opts) < 1 + 12:                                       # near-duplicates: one token of the truth changed
                x = a.clone()
                x[int(torch.randint(0, len(a), (1,), generator=g))] = int(torch.randint(V0, V1, (1,), generator=g))
                if not any(torch.equal(x, o) for o in opts):
                    opts.append(x); kind.append(1)
            while len(opts) < 1 + 12 + 40:                                  # other rounds' / dialogs' answers
                x = pool[int(torch.randint(0, len(pool), (1,), generator=g))]
                if not any(torch.equal(x, o) for o in opts):
                    opts.append(x.clone()); kind.append(0)
            while len(opts) < G:                                            # random answers, visdial-like lengths
                x = torch.randint(V0, V1, (int(torch.randint(1, 5, (1,), generator=g)),), generator=g)
                if not any(torch.equal(x, o) for o in opts):
                    opts.append(x); kind.append(0)
            order = torch.randperm(G, generator=g).tolist()
            per_round.append([(opts[j], kind[j]) for j in order])
        slots.append(per_round)
    return g, dialogs, round_id, slots


def materialize(dialogs, round_id, slots):
    B, NR, G = len(slots), len(slots[0]), len(slots[0][0])
    ids = torch.zeros(B, NR, T, dtype=torch.long); seg = torch.zeros(B, NR, T, dtype=torch.long)
    dec = torch.zeros(B, NR, G, U, dtype=torch.long); datt = torch.zeros(B, NR, G, U)
    feats = torch.zeros(B, R, FD); loc = torch.zeros(B, R, 5)
    gt = torch.zeros(B, NR, dtype=torch.long)
    rel = torch.zeros(B, G)
    for b, d in enumerate(dialogs):
        feats[b], loc[b] = d["feats"], d["loc"]
        for r in range(NR):
            ids[b, r], seg[b, r] = context(d, r)
            a = d["ans"][r]
            for slot, (x, k) in enumerate(slots[b][r]):
                dec[b, r, slot], datt[b, r, slot], _ = dec_rows(x)
                if k == 2:
                    gt[b, r] = slot
                if r == int(round_id[b, 0]) - 1:                           # dense relevance of the annotated round
                    rel[b, slot] = 1.0 if k == 2 else 0.5 if k == 1 else (0.2 if len(x) == len(a) and bool(x[0] == a[0]) else 0.0)
    return dict(enc_input_ids=ids, enc_segments=seg, dec_input_ids=dec, dec_att_mask=datt, enc_image_feat=feats, enc_image_loc=loc,
                enc_image_mask=torch.ones(B, R), gt_option_inds=gt, gt_relevance=rel, round_id=round_id)


def separate_ties(model, params, g, dialogs, round_id, slots, min_gap=2e-3, rounds=30):
    """Candidates whose REFERENCE scores lie closer than `min_gap` would make "ranks bit-exact" a statement about fp32 summation
    order, not about the model: one member of every such pair (never the ground truth) is replaced by a fresh random answer and
    the set is scored again, until the smallest gap between neighbouring candidates is >= min_gap.  (bf16 noise on a score is
    1e-2 .. 1e-1: the close calls that flip under bf16 -- gaps of 2e-3 .. 1e-1 -- all stay in.)"""
    for it in range(rounds):
        ev = materialize(dialogs, round_id, slots)
        scores = reference_scores(model, params, ev)
        srt, idx = scores.sort(-1)
        close = (srt[..., 1:] - srt[..., :-1]) < min_gap
        n = int(close.sum())
        print("  tie separation pass %d: %d neighbouring pairs closer than %.0e" % (it, n, min_gap), flush=True)
        if n == 0:
            return ev, scores
        for b, r, j in close.nonzero().tolist():
            cand = [int(idx[b, r, j]), int(idx[b, r, j + 1])]
            cand.sort(key=lambda sl: slots[b][r][sl][1])                    # replace the lowest 'kind' of the two (never kind 2)
            sl = cand[0]
            if slots[b][r][sl][1] == 2:
                sl = cand[1]
            while True:
                x = torch.randint(V0, V1, (int(torch.randint(1, 5, (1,), generator=g)),), generator=g)
                if not any(torch.equal(x, o) for o, _ in slots[b][r]):
                    break
            slots[b][r][sl] = (x, 0)
    raise RuntimeError("ties not separated")


def reference_scores(model, params, ev, chunk=500):
    """evaluate_gen.py:45-106: every [dialog, round, option] row, 500-row chunks (evaluate_gen.py:29,76-106)."""
    B, NR, G, _ = ev["dec_input_ids"].shape
    n = B * NR * G
    ids = ev["enc_input_ids"][:, :, None].expand(B, NR, G, T).reshape(n, T)
    seg = ev["enc_segments"][:, :, None].expand(B, NR, G, T).reshape(n, T)
    feats = ev["enc_image_feat"][:, None, None].expand(B, NR, G, R, FD).reshape(n, R, FD)
    loc = ev["enc_image_loc"][:, None, None].expand(B, NR, G, R, 5).reshape(n, R, 5)
    imask = ev["enc_image_mask"][:, None, None].expand(B, NR, G, R).reshape(n, R)
    dec_ids = ev["dec_input_ids"].reshape(n, U)
    datt = ev["dec_att_mask"].reshape(n, U)
    out = []
    params["mode"] = "vd_eval_val"
    with torch.no_grad():
        for s in range(0, n, chunk):
            e = slice(s, s + chunk)
            inp = dict(enc_input_ids=ids[e], enc_segments=seg[e], enc_attention_mask=(ids[e] != 0).float(), enc_image_features=feats[e],
                       enc_image_spatials=loc[e], enc_image_mask=imask[e], dec_attention_mask=datt[e])
            d = dec_ids[e]
            _, lm = call_model(model, inp, d.clone(), None)
            lm = F.log_softmax(lm, dim=-1)
            tgt = d.new_zeros(d.shape)
            tgt[:, :-1] = d[:, 1:].clone()
            sc = torch.gather(lm, -1, tgt.unsqueeze(-1)).squeeze(-1)
            out.append((sc * (tgt != 0).float()).sum(-1))
    params["mode"] = "vd_train"
    return torch.cat(out).view(B, NR, G)


def main():
    cfg_dir = os.path.join(OUT, "_cfg")
    e_path, d_path = rh.write_tiny_configs(cfg_dir)
    model, params = rh.build_reference_model(e_path, d_path, mode="vd_train", seed=11)
    ck = os.path.join(OUT, "tiny_state_trained.npz")
    if os.environ.get("R4_REUSE_STATE") and os.path.exists(ck):          # iterate on the eval set without re-training
        with np.load(ck) as z:
            model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    else:
        train_checkpoint(model, params)
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    np.savez(ck, **npy(sd))
    du, vm, ou = rh.reference_utils()
    g, dialogs, round_id, slots = eval_slots()
    ev, scores = separate_ties(model, params, g, dialogs, round_id, slots)
    sp = vm.SparseGTMetrics()
    sp.observe(scores.clone(), ev["gt_option_inds"])
    spm = sp.retrieve(reset=True)
    nd = vm.NDCG()
    rid = ev["round_id"].squeeze(1)
    nd.observe(scores[torch.arange(scores.size(0)), rid - 1, :].clone(), ev["gt_relevance"])
    ndm = nd.retrieve(reset=True)
    ranks = vm.scores_to_ranks(scores.clone())
    srt = scores.sort(-1)[0]
    gaps = srt[..., 1:] - srt[..., :-1]
    out = {"in::" + k: (v.to(torch.int16) if v.dtype == torch.long and k.startswith(("enc_input", "enc_seg", "dec_input")) else v)
           for k, v in ev.items()}
    out.update(scores=scores, ranks=ranks.to(torch.int16), sparse=np.array([spm[k] for k in ("r@1", "r@5", "r@10", "mean", "mrr")]),
               ndcg=np.array([ndm["ndcg"]]), min_score_gap=gaps.min(), frac_gaps_below_1e2=(gaps < 1e-2).float().mean())
    np.savez_compressed(os.path.join(OUT, "tiny_evalset100.npz"), **npy(out))
    print("evalset100: r@1 %.4f r@5 %.4f r@10 %.4f mean %.3f mrr %.4f ndcg %.4f | score std within a round %.2f, min gap %.2e, "
          "gaps < 1e-2: %.3f" % (spm["r@1"], spm["r@5"], spm["r@10"], spm["mean"], spm["mrr"], ndm["ndcg"],
                                 scores.std(-1).mean().item(), gaps.min().item(), (gaps < 1e-2).float().mean().item()))
    import shutil
    shutil.rmtree(cfg_dir)
    for fn in ("tiny_state_trained.npz", "tiny_evalset100.npz"):
        print("  %-28s %8d bytes" % (fn, os.path.getsize(os.path.join(OUT, fn))))


if __name__ == "__main__":
    main()
