"""Dialog generation round: the build's counterpart of the inner loop of generate.py:122-228.

The reference's `generate.py` works unchanged with the drop-in classes (it only calls `model(**kwargs)` and flips
`params['mode']`).  This module restates the loop's host-side pieces so that they stop being per-row Python:

  * `append_to_context` -- write the sampled question / answer behind each row's current context (generate.py:145-158,
    203-218) as one vectorised scatter on whatever device the ids live on; same results as the reference's per-row
    slice assignment, including its overflow rule (a row whose context would exceed the sequence length receives a
    single [SEP] instead and is reported as abnormal);
  * `answer_perplexity` -- the "kind of trick" of generate.py:176-201: score the sampled answer with the answerer itself
    (labels = ids shifted left, [SEP] -> [PAD] in place, per-token loss) and return exp(sum / answer length);
  * `dialog_round` -- question sampling, context update, answer sampling, perplexity, context + segment update.
"""
import torch


def append_to_context(ctx_ids, ctx_len, new_ids, sep_id, segments=None, segment_value=None):
    """In place: ctx_ids[b, ctx_len[b] : ctx_len[b] + n_b] = new_ids[b, :n_b] with n_b = #non-zero tokens of row b; rows that
    would run past the end get one [SEP] at ctx_len[b] instead (n_b := 1).  `segments`, if given, receives `segment_value`
    over the same span.  ctx_len is advanced.  Returns (n, abnormal_rows)."""
    B, T = ctx_ids.shape
    U = new_ids.shape[1]
    dev = ctx_ids.device
    n = (new_ids != 0).sum(-1)
    start = ctx_len.clone()
    ok = (start + n) <= T
    if bool(((~ok) & (start >= T)).any()):
        raise RuntimeError("context already full: cannot place [SEP]")          # the reference raises here as well
    col = torch.arange(U, device=dev)[None, :]
    rows = torch.arange(B, device=dev)[:, None].expand(B, U)
    take = (col < n[:, None]) & ok[:, None]
    pos = start[:, None] + col
    ctx_ids[rows[take], pos[take]] = new_ids[take]
    bad = ~ok
    if bool(bad.any()):
        ctx_ids[bad.nonzero().squeeze(1), start[bad]] = sep_id
    n_eff = torch.where(ok, n, torch.ones_like(n))
    if segments is not None:
        span = col < n_eff[:, None]
        segments[rows[span], pos[span]] = segment_value
    ctx_len += n_eff
    return n_eff, bad.nonzero().squeeze(1)


@torch.no_grad()
def answer_perplexity(model, enc_kwargs, ans_ids, reuse_decode_state=False):
    """generate.py:176-201.  `ans_ids` is mutated like the reference mutates it ([SEP] -> [PAD]); returns (ppl, ans_len).
    reuse_decode_state=True: `ans_ids` was just sampled by `model` from exactly `enc_kwargs` (nothing ran on the model in
    between) -- the teacher-forced pass then reuses that call's encoder states and cross-attention K/V
    (Engine.rescore_sampled) instead of running the encoder a second time, which is what the reference does."""
    if reuse_decode_state:
        loss, _ = model.engine.rescore_sampled(ans_ids, (ans_ids != 0).float(), loss_reduction=False)
    else:
        params = model.params
        mode = params["mode"]
        params["mode"] = "train"
        try:
            loss, _ = model(dec_input_ids=ans_ids, dec_attention_mask=(ans_ids != 0).float(), loss_reduction=False, **enc_kwargs)
        finally:
            params["mode"] = mode
    ans_len = (ans_ids != 0).sum(-1)
    loss = loss.reshape(ans_ids.shape[0], ans_ids.shape[1]).sum(-1) / ans_len
    return torch.exp(loss), ans_len


@torch.no_grad()
def dialog_round(q_model, a_model, state, sep_id=102, q_kwargs=None, a_kwargs=None):
    """One question / answer round.  `state`: dict with enc_image_features, enc_image_spatials, enc_image_mask,
    enc_input_ids, enc_segments, enc_input_len, dec_input_ids, dec_attention_mask (tensors on the model's device; the text
    tensors are updated in place).  Returns (ques_ids, ans_ids, ppl, abnormal_rows)."""
    q_kwargs = dict(temperature=0.7, top_k=7, top_p=0.0, ngram_blocking_size=4) if q_kwargs is None else q_kwargs
    a_kwargs = dict(temperature=0.7, top_k=7, top_p=0.0, ngram_blocking_size=0) if a_kwargs is None else a_kwargs

    def enc():
        return dict(enc_image_features=state["enc_image_features"], enc_image_spatials=state["enc_image_spatials"],
                    enc_image_mask=state["enc_image_mask"], enc_input_ids=state["enc_input_ids"],
                    enc_segments=state["enc_segments"], enc_attention_mask=(state["enc_input_ids"] != 0).float())

    ques = q_model(dec_input_ids=state["dec_input_ids"], dec_attention_mask=state["dec_attention_mask"], **q_kwargs, **enc())
    _, bad_q = append_to_context(state["enc_input_ids"], state["enc_input_len"], ques, sep_id)
    ans = a_model(dec_input_ids=state["dec_input_ids"], dec_attention_mask=state["dec_attention_mask"], **a_kwargs, **enc())
    ppl, _ = answer_perplexity(a_model, enc(), ans, reuse_decode_state=True)
    _, bad_a = append_to_context(state["enc_input_ids"], state["enc_input_len"], ans, sep_id,
                                 segments=state["enc_segments"], segment_value=1)
    return ques, ans, ppl, torch.cat((bad_q, bad_a))
