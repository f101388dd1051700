"""Step driver: the build's counterpart of `train_gen.forward` (train_gen.py:29-136).

Same batch contract (a dict of per-dialog tensors stacked on a leading batch dimension),
same row semantics: flatten [B, rounds, 1, L] -> [B*rounds, L]; in train mode keep rows whose label row is
not all zero and draw `batch_size` of them with replacement (torch.multinomial on the host RNG, exactly like
the reference); row r of a dialog uses the image tensors of dialog r // rounds.  Differences that do not
change results: image tensors may be passed UNEXPANDED ([B,37,2048] instead of the 10x host expansion of
train_gen.py:311-321) and the dead inputs (image_target / image_label / mlm labels / sep indices / hist len)
are neither gathered nor copied to the device.
"""
import torch

_TEXT_KEYS = ("enc_input_ids", "enc_segments", "enc_att_mask", "dec_input_ids", "dec_att_mask")


def select_rows(batch, params, sample_indices=None, generator=None):
    """Pure index work (device agnostic, bit-exact): returns (rows dict, sample_indices)."""
    ids = batch["enc_input_ids"]
    rows_per_dialog = 1
    for s in ids.shape[1:-1]:
        rows_per_dialog *= s
    flat = {k: batch[k].reshape(-1, batch[k].shape[-1]) for k in _TEXT_KEYS}
    n_rows = flat["enc_input_ids"].shape[0]
    train = "train" in params["mode"]
    if train:
        labels = batch["dec_labels"].reshape(-1, batch["dec_labels"].shape[-1])
        if sample_indices is None:
            cand = (labels.sum(-1) != 0).float()                                   # train_gen.py:66
            sample_indices = torch.multinomial(cand, params["batch_size"], replacement=True, generator=generator)
    elif sample_indices is None:
        sample_indices = torch.arange(n_rows)
    out = {k: v[sample_indices] for k, v in flat.items()}
    if train:
        out["dec_labels"] = labels[sample_indices]
    for k, tail in (("enc_image_feat", 2), ("enc_image_loc", 2), ("enc_image_mask", 1)):
        v = batch[k]
        if v.dim() == tail + 1:                       # unexpanded [B, ...]: row r -> dialog r // rows_per_dialog
            out[k] = v[torch.div(sample_indices, rows_per_dialog, rounding_mode="floor")]
        else:                                         # reference layout [B, rounds, 1, ...]
            out[k] = v.reshape((-1,) + tuple(v.shape[-tail:]))[sample_indices]
    return out, sample_indices


def forward(model, batch, params, sample_indices=None, generator=None):
    """lm_loss, lm_scores = forward(model, batch, params)  -- drop-in for train_gen.forward."""
    rows, _ = select_rows(batch, params, sample_indices, generator)
    dev = params["device"]
    rows = {k: v.to(dev, non_blocking=True) for k, v in rows.items()}
    loss, scores = model(
        enc_image_features=rows["enc_image_feat"], enc_image_spatials=rows["enc_image_loc"],
        enc_image_mask=rows["enc_image_mask"], enc_input_ids=rows["enc_input_ids"], enc_segments=rows["enc_segments"],
        enc_attention_mask=rows["enc_att_mask"], dec_input_ids=rows["dec_input_ids"],
        dec_attention_mask=rows["dec_att_mask"], dec_labels=rows.get("dec_labels"))
    if "train" in params["mode"]:
        loss = loss.mean()
    return loss, scores
