"""Token-id side of the sampling decode (utils/decoding_utils.py:4-77, models/visual_dialog_model.py:113-119).

Integer / index work that must be bit-exact with the reference; it runs as torch index plumbing on whatever
device the logits live on (the n-gram table is built on the host from one transfer per call, like the
reference's `.tolist()` loops but without the per-row synchronisation)."""
import torch
import torch.nn.functional as F

NEG_INF = -float("inf")


def batch_top_k_top_p_sampling(logits, top_k=0, top_p=0.0, filter_value=NEG_INF):
    assert logits.dim() == 2
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, filter_value)
    if top_p > 0.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = False
        logits = logits.masked_fill(remove.gather(-1, sorted_idx.argsort(-1)), filter_value)
    return logits


def batch_ngram_blocking(logits, enc_input_ids, dec_input_ids, ngram_size=0, filter_value=NEG_INF,
                         special_token_ids=(0, 100, 101, 102, 103)):
    """Ban every token that would complete an n-gram already present in `enc_input_ids` (n-grams that touch a
    special token are ignored)."""
    assert logits.dim() == 2
    if ngram_size <= 0:
        return logits
    special = set(special_token_ids)
    hist = enc_input_ids.tolist()
    dec = dec_input_ids.tolist()
    cur = dec_input_ids.shape[-1]
    start = cur + 1 - ngram_size
    rows, cols = [], []
    for b, toks in enumerate(hist):
        prefix = tuple(dec[b][start:cur])          # python slice semantics, as in the reference
        if len(prefix) != ngram_size - 1:
            continue
        for s in range(len(toks) - ngram_size + 1):
            gram = toks[s:s + ngram_size]
            if tuple(gram[:-1]) == prefix and not (special & set(gram)):
                rows.append(b)
                cols.append(gram[-1])
    if rows:
        logits = logits.clone()
        logits[torch.tensor(rows, device=logits.device), torch.tensor(cols, device=logits.device)] = filter_value
    return logits


def pad_after_eos(sequence, eos_token_id, pad_token_id):
    eq = (sequence == eos_token_id).long()
    after = (torch.cumsum(eq, dim=1) - eq) > 0
    return sequence.masked_fill(after, pad_token_id)
