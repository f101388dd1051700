"""Retrieval metrics of the generative evaluation (utils/visdial_metrics.py:21-195): integer rank work on the host,
bit-exact with the reference (ties resolved by torch.sort exactly as there)."""
import torch


def scores_to_ranks(scores):
    """[B, rounds, options] scores -> 1-based rank of every option (descending score)."""
    b, r, o = scores.shape
    order = scores.reshape(-1, o).sort(1, descending=True)[1]
    ranks = torch.empty_like(order)
    ranks.scatter_(1, order, torch.arange(o, device=order.device).expand_as(order))
    return (ranks + 1).view(b, r, o)


class SparseGTMetrics(object):
    """R@1/5/10, mean rank, MRR against one ground-truth option per round (visdial_metrics.py:41-117)."""

    def __init__(self):
        self._ranks = []

    def observe(self, predicted_scores, target_ranks):
        ranks = scores_to_ranks(predicted_scores.detach())
        b, r, o = ranks.shape
        flat = ranks.view(b * r, o)
        gt = target_ranks.reshape(b * r).long().to(flat.device)
        self._ranks.extend(flat[torch.arange(b * r, device=flat.device), gt].cpu().tolist())

    def retrieve(self, reset=True):
        out = {}
        if self._ranks:
            r = torch.tensor(self._ranks).float()
            out = {"r@1": (r <= 1).float().mean().item(), "r@5": (r <= 5).float().mean().item(),
                   "r@10": (r <= 10).float().mean().item(), "mean": r.mean().item(), "mrr": r.reciprocal().mean().item()}
        if reset:
            self.reset()
        return out

    def reset(self):
        self._ranks = []


class NDCG(object):
    """visdial_metrics.py:119-195."""

    def __init__(self):
        self._num, self._den = 0.0, 0.0

    def observe(self, predicted_scores, target_relevance):
        ranks = scores_to_ranks(predicted_scores.detach().unsqueeze(1)).squeeze(1)
        rel = target_relevance.to(ranks.device)
        k = (rel != 0).sum(-1)
        rankings = torch.sort(ranks, dim=-1)[1]
        best = torch.sort(rel, dim=-1, descending=True)[1]
        for b in range(ranks.shape[0]):
            n = int(k[b])
            disc = torch.log2(torch.arange(n).float() + 2)
            dcg = (rel[b][rankings[b][:n]].cpu().float() / disc).sum()
            ideal = (rel[b][best[b][:n]].cpu().float() / disc).sum()
            self._num += float(dcg / ideal)
        self._den += ranks.shape[0]

    def retrieve(self, reset=True):
        out = {"ndcg": float(self._num / self._den)} if self._den > 0 else {}
        if reset:
            self.reset()
        return out

    def reset(self):
        self._num, self._den = 0.0, 0.0
