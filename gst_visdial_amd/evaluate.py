"""Generative evaluation: the build's counterpart of evaluate_gen.evaluate (evaluate_gen.py:22-142).

Same batch contract (per-dialog tensors [B, rounds, options, L] from the eval dataloader) and the
same metrics; the model work differs in one way that does not change results: every (dialog, round) context is encoded
ONCE and shared by its `options` answer candidates instead of being re-encoded per candidate."""
import torch

from .metrics import SparseGTMetrics, NDCG, scores_to_ranks


def score_batch(model, batch, device, rounds_per_call=20):
    """-> scores [B, rounds, options] (sum of target-token log-probabilities of every candidate answer)."""
    ids = batch["enc_input_ids"]
    B, R_, O = ids.shape[0], ids.shape[1], ids.shape[2]
    T, U = ids.shape[-1], batch["dec_input_ids"].shape[-1]
    seg, att = batch["enc_segments"], batch["enc_att_mask"]
    # the `options` rows of a round normally carry the same context (dataloader_visdial_gen.py:379-388 encodes it once per
    # option): then option 0 is the round's encoder input.  Checked, not assumed -- e.g. attack='random_token' masks each
    # option's copy differently -- and otherwise every option row is scored against its own context (group = 1)
    if not (bool((ids == ids[:, :, :1]).all()) and bool((seg == seg[:, :, :1]).all()) and bool((att == att[:, :, :1]).all())):
        return _score_batch_per_row(model, batch, device, rounds_per_call)
    enc_ids = ids[:, :, 0].reshape(B * R_, T)
    enc_seg = batch["enc_segments"][:, :, 0].reshape(B * R_, T)
    enc_att = batch["enc_att_mask"][:, :, 0].reshape(B * R_, T)
    feat, loc, imask = batch["enc_image_feat"], batch["enc_image_loc"], batch["enc_image_mask"]
    dial = torch.arange(B).repeat_interleave(R_)
    dec_ids = batch["dec_input_ids"].reshape(B * R_, O, U)
    dec_att = batch["dec_att_mask"].reshape(B * R_, O, U)
    out = []
    for s in range(0, B * R_, rounds_per_call):
        e = slice(s, min(s + rounds_per_call, B * R_))
        d = dial[e]
        sc = model.score_candidates(feat[d].to(device), loc[d].to(device), imask[d].to(device), enc_ids[e].to(device),
                                    enc_seg[e].to(device), enc_att[e].to(device), dec_ids[e].reshape(-1, U).to(device),
                                    dec_att[e].reshape(-1, U).to(device), O)
        out.append(sc)
    return torch.cat(out).view(B, R_, O)


def _score_batch_per_row(model, batch, device, rounds_per_call):
    """evaluate_gen.py:45-106 literally: every [dialog, round, option] row through encoder and decoder."""
    ids = batch["enc_input_ids"]
    B, R_, O, T = ids.shape
    U = batch["dec_input_ids"].shape[-1]
    n = B * R_ * O
    flat = lambda k, L: batch[k].reshape(n, L)
    dial = torch.arange(B).repeat_interleave(R_ * O)
    out = []
    rows = max(1, rounds_per_call) * O
    for s in range(0, n, rows):
        e = slice(s, min(s + rows, n))
        d = dial[e]
        out.append(model.score_candidates(batch["enc_image_feat"][d].to(device), batch["enc_image_loc"][d].to(device),
                                          batch["enc_image_mask"][d].to(device), flat("enc_input_ids", T)[e].to(device),
                                          flat("enc_segments", T)[e].to(device), flat("enc_att_mask", T)[e].to(device),
                                          flat("dec_input_ids", U)[e].to(device), flat("dec_att_mask", U)[e].to(device), 1))
    return torch.cat(out).view(B, R_, O)


@torch.no_grad()
def evaluate(model, dataloader, params, mode="vd_eval_val", scorer=None, group=None):
    """Returns (ranks_json, metrics) like evaluate_gen.evaluate + its logged metric dict.

    Multi-GPU: with an initialised process group the dialogs are sharded by batch index (batch i goes to
    rank i % world), there is no exchange on the data path, and the metric state -- the ground-truth ranks, the NDCG
    sums, the ranks json -- is all-gathered once at the end, so every rank returns the metrics of the whole set.
    `scorer(model, batch, device) -> [B, rounds, options]` defaults to `score_batch`."""
    import torch.distributed as dist
    scorer = scorer or score_batch
    sparse, ndcg, ranks_json = SparseGTMetrics(), NDCG(), []
    if hasattr(model, "eval"):
        model.eval()
    device = params["device"]
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if sharded else (0, 1)
    for i, batch in enumerate(dataloader):
        if i % world != rank:
            continue
        scores = scorer(model, batch, device)
        if mode == "vd_eval_val":
            sparse.observe(scores, batch["gt_option_inds"])
            if params.get("vd_version", "1.0") == "1.0" and "gt_relevance" in batch:
                rid = batch["round_id"].squeeze(1)
                ndcg.observe(scores[torch.arange(scores.size(0)), rid - 1, :], batch["gt_relevance"])
        else:
            ranks = scores_to_ranks(scores).squeeze(1)
            for j in range(scores.shape[0]):
                ranks_json.append({"image_id": batch["image_id"][j].item(), "round_id": int(batch["round_id"][j].item()),
                                   "ranks": [r.item() for r in ranks[j][:]]})
    if sharded:
        parts = [None] * world
        dist.all_gather_object(parts, (sparse._ranks, ndcg._num, ndcg._den, ranks_json), group=group)
        sparse._ranks = [r for p in parts for r in p[0]]
        ndcg._num, ndcg._den = sum(p[1] for p in parts), sum(p[2] for p in parts)
        ranks_json = [e for p in parts for e in p[3]]
    metrics = {}
    if mode == "vd_eval_val":
        metrics.update(sparse.retrieve(reset=True))
        metrics.update(ndcg.retrieve(reset=True))
    return ranks_json, metrics
