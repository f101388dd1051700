"""Token-id side of the sampling decode (utils/decoding_utils.py:4-77, models/visual_dialog_model.py:113-119).

Integer / index work that must be bit-exact with the reference; it runs as torch index plumbing on whatever
device the logits live on, without host round trips (the reference's n-gram ban walks Python dictionaries built from
`.tolist()` every step)."""
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


def ngram_banned_mask(enc_input_ids, dec_input_ids, ngram_size, vocab, device=None, special_token_ids=(0, 100, 101, 102, 103)):
    """bool [B, vocab + 1] (column `vocab` is a dummy): True for every token that would complete an n-gram already present in
    `enc_input_ids` (n-grams that touch a special token are ignored) -- or None when nothing can be banned.  Vectorised on the
    device, free of host synchronisation (scatter instead of boolean-mask indexing, scalar compares instead of an uploaded
    table): capturable into a hipGraph; same bans as the reference's per-row dictionary loops (`_ngram_blocking_loop`)."""
    cur = dec_input_ids.shape[-1]
    T = enc_input_ids.shape[-1]
    n = ngram_size
    if n <= 0 or cur < n - 1 or T < n:      # (python slice semantics of the reference: a short prefix never matches)
        return None
    dev = device if device is not None else enc_input_ids.device
    hist = enc_input_ids.to(dev)
    win = hist.unfold(1, n, 1)                                              # [B, T-n+1, n] every n-gram of the history
    bad = torch.zeros_like(win, dtype=torch.bool)
    for sid in special_token_ids:
        bad |= win == sid
    clean = ~bad.any(-1)                                                    # n-grams without a special token
    if n > 1:
        prefix = dec_input_ids.to(dev)[:, cur - (n - 1):cur]                # the last n-1 generated tokens
        hit = (win[..., :n - 1] == prefix[:, None, :]).all(-1) & clean
    else:
        hit = clean
    last_tok = win[..., n - 1]
    idx = torch.where(hit, last_tok, torch.full_like(last_tok, vocab))      # n-grams that do not hit point at the dummy column
    return torch.zeros(hist.shape[0], vocab + 1, dtype=torch.bool, device=dev).scatter_(1, idx, True)


def batch_ngram_blocking(logits, enc_input_ids, dec_input_ids, ngram_size=0, filter_value=NEG_INF,
                         special_token_ids=(0, 100, 101, 102, 103)):
    """utils/decoding_utils.py:34-77 on the logits' device (see `ngram_banned_mask`)."""
    assert logits.dim() == 2
    V = logits.shape[-1]
    banned = ngram_banned_mask(enc_input_ids, dec_input_ids, ngram_size, V, logits.device, special_token_ids)
    if banned is None:
        return logits
    return logits.masked_fill(banned[:, :V], filter_value)


def _ngram_blocking_loop(logits, enc_input_ids, dec_input_ids, ngram_size=0, filter_value=NEG_INF,
                         special_token_ids=(0, 100, 101, 102, 103)):
    """Per-row restatement of utils/decoding_utils.py:34-77 (host loops); kept as the checker of the vectorised form."""
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


def draw_from_uniform(prob, u):
    """Inverse-CDF draw: token = first index whose cumulative probability reaches u * total, u [B] in (0, 1).
    The stand-in for torch.multinomial(prob, 1) (models/visual_dialog_model.py:104-108) when the caller supplies the
    randomness: multinomial's stream differs between devices, this rule does not, so sampled ids can be compared with the
    reference run under the same uniforms (oracle/make_golden_r2.py patches the reference's multinomial with this rule)."""
    c = torch.cumsum(prob.float(), dim=-1)
    x = u.to(c.device, torch.float32).reshape(-1, 1) * c[:, -1:]
    idx = (c < x).sum(-1, keepdim=True)
    return idx.clamp_(max=prob.shape[-1] - 1)
