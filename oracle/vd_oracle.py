"""CPU oracle: a plain-PyTorch fp32 restatement of the gst-visdial enc_dec_a hot path.

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file; the product
(`gst_visdial_amd`) never does and fails loudly without its HIP library.

What it restates (reference = gicheonkang/gst-visdial, paths relative to
/root/reference; the decoder stack is third-party `transformers==4.16.2`
`modeling_bert.BertEncoder`, call sites models/visual_dialog_decoder.py:15,203,
300-311 -- its published algorithm is restated in `_decoder_layer`):

  * the two-stream ViLBERT encoder          models/vilbert_dialog.py:298-912,1325-1427
  * VLFusion                                models/visual_dialog_model.py:123-135
  * the BERT-generation decoder + LM head   models/visual_dialog_decoder.py:33-86,219-339
  * the enc-dec glue / sampling decode      models/visual_dialog_model.py:24-120
  * row sampling of the train step driver   train_gen.py:29-136
  * candidate scoring of evaluate_gen       evaluate_gen.py:94-106
  * decoding filters                        utils/decoding_utils.py:4-77
  * retrieval metrics                       utils/visdial_metrics.py:21-195
  * LR schedule                             utils/optim_utils.py:8-26

Pinning: every function here is checked (tests/test_oracle_golden.py) against
golden vectors under tests/golden/ that `oracle/make_golden.py` produced by
importing and running the reference itself in the build container, and
(tests/test_oracle_vs_reference.py, container only) against the live reference
at the full bert-base configuration.

The oracle is written functionally over the reference's own `state_dict()`
(key names are the checkpoint layout of train_gen.py:346-351) so a reference
checkpoint drives it directly.
"""
import math

import torch
import torch.nn.functional as F

ENC = "encoder.bert_pretrained.bert."
DEC = "decoder.decoder.bert."
LMH = "decoder.decoder.lm_head."


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def gelu_erf(x):
    """models/vilbert_dialog.py:115-121 (exact erf GELU; HF 'gelu' is the same function)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm_tf(x, w, b, eps=1e-12):
    """models/vilbert_dialog.py:283-296: biased variance, epsilon inside the sqrt.
    `nn.LayerNorm(eps=1e-12)` of the decoder stack is the same arithmetic."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


class DropMasks(object):
    """Injected dropout masks (test hook).  `train` arguments throughout this file are either a bool (True: `F.dropout` on the
    host RNG, what the reference's nn.Dropout modules do) or one of these: a table  site label -> keep mask  (bool / 0-1
    tensor of the dropped tensor's shape).  A site then computes  x * keep / (1 - p)  -- nn.Dropout's arithmetic with the
    Bernoulli draw replaced by the given mask and the scale taken from THIS file's own `p` for the site (the reference's
    config value), so a product that applies the right mask with the wrong probability / scale at a site does not match.

    Site labels (every nn.Dropout the enc_dec_a train step executes; reference lines in the functions that use them):
      emb.enc / emb.dec            BertEmbeddingsDialog.dropout, encoder / decoder call          vilbert_dialog.py:352
      vemb                         BertImageEmbeddings.dropout                                   vilbert_dialog.py:1427
      t<i>.attn .ln1 .ln2          BertLayer i: attention probs, attention output, FFN output    :401,419,461
      v<i>.attn .ln1 .ln2          BertImageLayer i: the same three                              :528,546,588
      c<i>.attn1 .attn2            BertBiAttention: dropout1 (probs over vision keys), dropout2  :707,728
      c<i>.ln1 .ln2 .vln .tln      BertBiOutput.dropout1 / dropout2, v_output / t_output         :735-742,588,461
      vlf.v / vlf.l                VLFusion.dropout on the vision / text rows of the concat      visual_dialog_model.py:133
      d<i>.attn .xattn .ln1 .ln2 .ln3   decoder layer i (transformers 4.16.2 BertLayer with cross-attention)
    `used` records the labels a forward pass asked for, `p_used` the probability the oracle applied there."""

    def __init__(self, table):
        self.table, self.used, self.p_used = table, [], {}

    def keep(self, label, shape, p):
        if label not in self.table:
            raise KeyError("no injected dropout mask for site %r" % (label,))
        if label in self.p_used:
            raise KeyError("dropout site %r asked for twice" % (label,))
        self.used.append(label)
        self.p_used[label] = p
        m = self.table[label]
        return (m != 0).reshape(shape).to(torch.float32)


def _drop(x, p, train, label=None):
    """nn.Dropout(p) in training mode (every call site cites the reference module it stands for)."""
    if not train or p <= 0:
        return x
    if isinstance(train, DropMasks):
        return x * (train.keep(label, x.shape, p) * (1.0 / (1.0 - p)))
    return F.dropout(x, p, training=True)


def _heads(x, nh):
    B, L, H = x.shape
    return x.view(B, L, nh, H // nh).permute(0, 2, 1, 3)


def _merge(x):
    B, nh, L, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, nh * d)


def attention_core(q, k, v, add_mask, nh, p_attn, train, label=None):
    """softmax(q k^T / sqrt(d) + mask) v -- models/vilbert_dialog.py:389-405
    (scale applied to the scores *before* the additive mask)."""
    qh, kh, vh = _heads(q, nh), _heads(k, nh), _heads(v, nh)
    scores = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(qh.shape[-1])
    scores = scores + add_mask
    probs = torch.softmax(scores, dim=-1)
    probs = _drop(probs, p_attn, train, label)
    return _merge(torch.matmul(probs, vh))


# ----------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------
def text_embeddings(sd, prefix, ids, token_type_ids, cfg, train, label="emb.enc"):
    """models/vilbert_dialog.py:324-352 (BertEmbeddingsDialog.forward).
    type embedding = token_type_embeddings[tt] if tt < type_vocab_size else
    token_type_embeddings_extension[tt - type_vocab_size]; `sep_indices`,
    `sep_embeddings` and the sinusoid table are unused."""
    T = ids.shape[1]
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(ids)
    tv = cfg["type_vocab_size"]
    words = F.embedding(ids, sd[prefix + "word_embeddings.weight"])
    pos = sd[prefix + "position_embeddings.weight"][:T].unsqueeze(0)
    ext = token_type_ids - tv
    ext_mask = (ext >= 0).float()
    base_mask = (token_type_ids < tv).float()
    ext_idx = (ext.float() * ext_mask).long()
    base_idx = (token_type_ids.float() * base_mask).long()
    tt = (F.embedding(base_idx, sd[prefix + "token_type_embeddings.weight"]) * base_mask.unsqueeze(-1)
          + F.embedding(ext_idx, sd[prefix + "token_type_embeddings_extension.weight"]) * ext_mask.unsqueeze(-1))
    e = layer_norm_tf(words + pos + tt, sd[prefix + "LayerNorm.weight"], sd[prefix + "LayerNorm.bias"])
    return _drop(e, cfg["hidden_dropout_prob"], train, label)


def image_embeddings(sd, feats, locs, cfg, train):
    """models/vilbert_dialog.py:1420-1427 (dropout uses hidden_dropout_prob, :1418)."""
    p = ENC + "v_embeddings."
    e = _lin(sd, p + "image_embeddings", feats) + _lin(sd, p + "image_location_embeddings", locs)
    e = layer_norm_tf(e, sd[p + "LayerNorm.weight"], sd[p + "LayerNorm.bias"])
    return _drop(e, cfg["hidden_dropout_prob"], train, "vemb")


def _self_block(sd, p, x, add_mask, nh, p_attn, p_hid, train, lab=None):
    """attention + output sublayer: vilbert_dialog.py:380-431 / 507-558."""
    ctx = attention_core(_lin(sd, p + "attention.self.query", x), _lin(sd, p + "attention.self.key", x),
                         _lin(sd, p + "attention.self.value", x), add_mask, nh, p_attn, train, "%s.attn" % lab)
    h = _drop(_lin(sd, p + "attention.output.dense", ctx), p_hid, train, "%s.ln1" % lab)
    return layer_norm_tf(h + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"])


def _ffn_block(sd, p_int, p_out, x, p_hid, train, label=None):
    """intermediate + output sublayer: vilbert_dialog.py:445-462 / 572-589."""
    inter = gelu_erf(_lin(sd, p_int + ".dense", x))
    h = _drop(_lin(sd, p_out + ".dense", inter), p_hid, train, label)
    return layer_norm_tf(h + x, sd[p_out + ".LayerNorm.weight"], sd[p_out + ".LayerNorm.bias"])


def text_layer(sd, i, x, add_mask, cfg, train):
    """BertLayer, models/vilbert_dialog.py:465-476."""
    p = ENC + "encoder.layer.%d." % i
    a = _self_block(sd, p, x, add_mask, cfg["num_attention_heads"],
                    cfg["attention_probs_dropout_prob"], cfg["hidden_dropout_prob"], train, "t%d" % i)
    return _ffn_block(sd, p + "intermediate", p + "output", a, cfg["hidden_dropout_prob"], train, "t%d.ln2" % i)


def vision_layer(sd, i, x, add_mask, cfg, train):
    """BertImageLayer, models/vilbert_dialog.py:592-603."""
    p = ENC + "encoder.v_layer.%d." % i
    a = _self_block(sd, p, x, add_mask, cfg["v_num_attention_heads"],
                    cfg["v_attention_probs_dropout_prob"], cfg["v_hidden_dropout_prob"], train, "v%d" % i)
    return _ffn_block(sd, p + "intermediate", p + "output", a, cfg["v_hidden_dropout_prob"], train, "v%d.ln2" % i)


def connection_layer(sd, i, xv, mask_v, xt, mask_t, cfg, train):
    """BertConnectionLayer, models/vilbert_dialog.py:646-773.
    stream 1 = vision, stream 2 = text.  ctx1 (text queries over vision keys) feeds the
    TEXT output branch and ctx2 (vision queries over text keys) the VISION branch
    (the swap at :765); q_dense1/2 are unused."""
    p = ENC + "encoder.c_layer.%d." % i
    nh = cfg["bi_num_attention_heads"]
    b = p + "biattention."
    q1, k1, v1 = (_lin(sd, b + n + "1", xv) for n in ("query", "key", "value"))
    q2, k2, v2 = (_lin(sd, b + n + "2", xt) for n in ("query", "key", "value"))
    ctx1 = attention_core(q2, k1, v1, mask_v, nh, cfg["v_attention_probs_dropout_prob"], train, "c%d.attn1" % i)  # [B,T,Hb]
    ctx2 = attention_core(q1, k2, v2, mask_t, nh, cfg["attention_probs_dropout_prob"], train, "c%d.attn2" % i)    # [B,R,Hb]
    o = p + "biOutput."
    hv = _drop(_lin(sd, o + "dense1", ctx2), cfg["v_hidden_dropout_prob"], train, "c%d.ln1" % i)
    ht = _drop(_lin(sd, o + "dense2", ctx1), cfg["hidden_dropout_prob"], train, "c%d.ln2" % i)
    av = layer_norm_tf(hv + xv, sd[o + "LayerNorm1.weight"], sd[o + "LayerNorm1.bias"])
    at = layer_norm_tf(ht + xt, sd[o + "LayerNorm2.weight"], sd[o + "LayerNorm2.bias"])
    ov = _ffn_block(sd, p + "v_intermediate", p + "v_output", av, cfg["v_hidden_dropout_prob"], train, "c%d.vln" % i)
    ot = _ffn_block(sd, p + "t_intermediate", p + "t_output", at, cfg["hidden_dropout_prob"], train, "c%d.tln" % i)
    return ov, ot


def encoder_schedule(cfg):
    """The order in which BertEncoder.forward (models/vilbert_dialog.py:831-905) runs its
    sublayers with fixed_*_layer=0: a list of ('t', i) / ('v', i) / ('c', i)."""
    order, vs, ts = [], 0, 0
    for c, (ve, te) in enumerate(zip(cfg["v_biattention_id"], cfg["t_biattention_id"])):
        order += [("v", i) for i in range(vs, ve)]
        order += [("t", i) for i in range(ts, te)]
        order.append(("c", c))
        vs, ts = ve, te
    order += [("v", i) for i in range(vs, cfg["v_num_hidden_layers"])]
    order += [("t", i) for i in range(ts, cfg["num_hidden_layers"])]
    return order


def encoder_forward(sd, cfg, ids, segments, att_mask, feats, locs, img_mask, train=False):
    """BertModel.forward, models/vilbert_dialog.py:1325-1407, for the enc_dec arch: returns
    (enc_hidden_t [B,T,H], enc_hidden_v [B,R,Hv]).  The pooler / cls heads of :1400-1401,1482
    are dead in enc_dec (outputs discarded at :1485-1487) and are not evaluated."""
    mask_t = (1.0 - att_mask.float())[:, None, None, :] * -10000.0
    mask_v = (1.0 - img_mask.float())[:, None, None, :] * -10000.0
    xt = text_embeddings(sd, ENC + "embeddings.", ids, segments, cfg, train)
    xv = image_embeddings(sd, feats, locs, cfg, train)
    for kind, i in encoder_schedule(cfg):
        if kind == "t":
            xt = text_layer(sd, i, xt, mask_t, cfg, train)
        elif kind == "v":
            xv = vision_layer(sd, i, xv, mask_v, cfg, train)
        else:
            xv, xt = connection_layer(sd, i, xv, mask_v, xt, mask_t, cfg, train)
    return xt, xv


def vl_fusion(sd, enc_t, enc_v, att_mask, img_mask, train=False):
    """models/visual_dialog_model.py:131-135: vision first, dropout 0.1."""
    hv, hl = _lin(sd, "vlfusion.fc_v", enc_v), _lin(sd, "vlfusion.fc_l", enc_t)
    mask = torch.cat((img_mask, att_mask), dim=1)
    if isinstance(train, DropMasks):      # element-wise: dropout of the concat == concat of the halves' dropouts
        return torch.cat((_drop(hv, 0.1, train, "vlf.v"), _drop(hl, 0.1, train, "vlf.l")), dim=1), mask
    return _drop(torch.cat((hv, hl), dim=1), 0.1, train), mask


# ----------------------------------------------------------------------------
# decoder (transformers 4.16.2 BertEncoder with is_decoder + add_cross_attention)
# ----------------------------------------------------------------------------
def decoder_masks(dec_att_mask, enc_mask, B, U):
    """models/visual_dialog_decoder.py:267-287 + transformers 4.16.2
    `get_extended_attention_mask` (causal x padding, (1-m)*-10000) and
    `invert_attention_mask` ((1-m)*-1e9 in fp32)."""
    if dec_att_mask is None:
        dec_att_mask = torch.ones(B, U)
    i = torch.arange(U)
    causal = (i[None, :] <= i[:, None]).float()                      # [U(q), U(k)]
    self_mask = causal[None, None] * dec_att_mask.float()[:, None, None, :]
    self_add = (1.0 - self_mask) * -10000.0
    cross_add = (1.0 - enc_mask.float())[:, None, None, :] * -1e9
    return self_add, cross_add


def _decoder_layer(sd, i, y, self_add, enc_h, cross_add, cfg, train):
    """transformers 4.16.2 modeling_bert.BertLayer.forward with cross-attention:
    self-attention -> post-LN -> cross-attention -> post-LN -> FFN -> post-LN."""
    p = DEC + "encoder.layer.%d." % i
    nh, pa, ph = cfg["num_attention_heads"], cfg["attention_probs_dropout_prob"], cfg["hidden_dropout_prob"]
    eps = cfg.get("layer_norm_eps", 1e-12)
    ctx = attention_core(_lin(sd, p + "attention.self.query", y), _lin(sd, p + "attention.self.key", y),
                         _lin(sd, p + "attention.self.value", y), self_add, nh, pa, train, "d%d.attn" % i)
    h = _drop(_lin(sd, p + "attention.output.dense", ctx), ph, train, "d%d.ln1" % i)
    y1 = layer_norm_tf(h + y, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
    ctx = attention_core(_lin(sd, p + "crossattention.self.query", y1), _lin(sd, p + "crossattention.self.key", enc_h),
                         _lin(sd, p + "crossattention.self.value", enc_h), cross_add, nh, pa, train, "d%d.xattn" % i)
    h = _drop(_lin(sd, p + "crossattention.output.dense", ctx), ph, train, "d%d.ln2" % i)
    y2 = layer_norm_tf(h + y1, sd[p + "crossattention.output.LayerNorm.weight"],
                       sd[p + "crossattention.output.LayerNorm.bias"], eps)
    inter = gelu_erf(_lin(sd, p + "intermediate.dense", y2))
    h = _drop(_lin(sd, p + "output.dense", inter), ph, train, "d%d.ln3" % i)
    return layer_norm_tf(h + y2, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def decoder_hidden(sd, dec_cfg, dec_ids, dec_att_mask, enc_h, enc_mask, train=False, emb_dropout_p=None):
    """BertGenerationEncoder.forward, models/visual_dialog_decoder.py:219-323.
    The embedding module is the encoder's (train_gen.py:293), called with segments = 0.  `emb_dropout_p`: the probability of
    that module's nn.Dropout -- the ENCODER config's hidden_dropout_prob (vilbert_dialog.py:321), whoever calls the module
    (pinned by tests/golden/tiny_train_dropout.npz, where the two configs differ); None = the decoder config's value."""
    B, U = dec_ids.shape
    self_add, cross_add = decoder_masks(dec_att_mask, enc_mask, B, U)
    emb_cfg = dec_cfg if emb_dropout_p is None else dict(dec_cfg, hidden_dropout_prob=emb_dropout_p)
    y = text_embeddings(sd, DEC + "embeddings.", dec_ids, None, emb_cfg, train, "emb.dec")
    for i in range(dec_cfg["num_hidden_layers"]):
        y = _decoder_layer(sd, i, y, self_add, enc_h, cross_add, dec_cfg, train)
    return y


def lm_logits(sd, y):
    """BertGenerationOnlyLMHead, models/visual_dialog_decoder.py:326-339 (own weight, aliased bias)."""
    return F.linear(y, sd[LMH + "decoder.weight"], sd[LMH + "bias"])


def shift_labels_(dec_ids, eos=102, pad=0):
    """models/visual_dialog_decoder.py:53-57: labels = ids shifted left; then IN PLACE
    replace eos by pad in the caller's `dec_ids`.  Returns labels."""
    labels = dec_ids.new_zeros(dec_ids.shape)
    labels[:, :-1] = dec_ids[:, 1:].clone()
    dec_ids.masked_fill_(dec_ids == eos, pad)
    return labels


def decoder_forward(sd, dec_cfg, dec_ids, dec_att_mask, enc_h, enc_mask, labels=None,
                    loss_reduction=True, train=False, want_loss=True, emb_dropout_p=None):
    """VisualDialogDecoder.forward, models/visual_dialog_decoder.py:33-86 -> (loss, logits, last_hidden)."""
    if labels is None and want_loss:
        labels = shift_labels_(dec_ids, dec_cfg.get("eos_token_id", 102), dec_cfg.get("pad_token_id", 0))
    y = decoder_hidden(sd, dec_cfg, dec_ids, dec_att_mask, enc_h, enc_mask, train, emb_dropout_p)
    logits = lm_logits(sd, y)
    loss = None
    if want_loss:
        V = logits.shape[-1]
        loss = F.cross_entropy(logits.reshape(-1, V), labels.reshape(-1), ignore_index=dec_cfg.get("pad_token_id", 0),
                               reduction="mean" if loss_reduction else "none")
    return loss, logits, y


def model_forward(sd, enc_cfg, dec_cfg, batch, train=False, loss_reduction=True):
    """EncoderDecoderModel.forward train/eval branch, models/visual_dialog_model.py:24-72.
    `batch` uses the keyword names of that signature.  Returns dict with loss, logits and
    the intermediate stages used as parity pins."""
    enc_t, enc_v = encoder_forward(sd, enc_cfg, batch["enc_input_ids"], batch["enc_segments"],
                                   batch["enc_attention_mask"], batch["enc_image_features"],
                                   batch["enc_image_spatials"], batch["enc_image_mask"], train)
    enc_h, enc_mask = vl_fusion(sd, enc_t, enc_v, batch["enc_attention_mask"], batch["enc_image_mask"], train)
    loss, logits, y = decoder_forward(sd, dec_cfg, batch["dec_input_ids"], batch.get("dec_attention_mask"),
                                      enc_h, enc_mask, batch.get("dec_labels"), loss_reduction, train,
                                      emb_dropout_p=enc_cfg["hidden_dropout_prob"])
    return dict(loss=loss, logits=logits, enc_hidden_t=enc_t, enc_hidden_v=enc_v,
                enc_hidden=enc_h, enc_mask=enc_mask, dec_hidden=y)


# ----------------------------------------------------------------------------
# decoding (models/visual_dialog_model.py:74-120, utils/decoding_utils.py)
# ----------------------------------------------------------------------------
def top_k_top_p_filter(logits, top_k=0, top_p=0.0, filter_value=-float("inf")):
    """utils/decoding_utils.py:4-35.  Mutates and returns `logits` like the reference."""
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < kth] = filter_value
    if top_p > 0.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        mask = remove.gather(-1, sorted_idx.argsort(-1))
        logits = logits.masked_fill(mask, filter_value)
    return logits


def ngram_block(logits, hist_ids, dec_ids, ngram_size=0, filter_value=-float("inf"),
                special=(0, 100, 101, 102, 103)):
    """utils/decoding_utils.py:38-77: ban every token that would complete an n-gram that
    occurs in `hist_ids` (n-grams touching a special token are skipped)."""
    if ngram_size <= 0:
        return logits
    special = set(special)
    cur = dec_ids.shape[-1]
    for b in range(hist_ids.shape[0]):
        toks = hist_ids[b].tolist()
        table = {}
        for s in range(len(toks) - ngram_size + 1):
            g = toks[s:s + ngram_size]
            if special & set(g):
                continue
            table.setdefault(tuple(g[:-1]), []).append(g[-1])
        start = cur + 1 - ngram_size
        # python slice semantics of the reference: a negative start wraps
        key = tuple(dec_ids[b][start:cur].tolist())
        banned = table.get(key, [])
        if banned:
            logits[b, banned] = filter_value
    return logits


def pad_after_eos(seq, eos=102, pad=0):
    """models/visual_dialog_model.py:113-119."""
    out = seq.clone()
    for b in range(out.shape[0]):
        hits = (out[b] == eos).nonzero()
        if hits.numel():
            out[b, hits[0, 0] + 1:] = pad
    return out


def sampling_decode(sd, enc_cfg, dec_cfg, batch, temperature, top_k, top_p, ngram_blocking_size,
                    max_len=18, draw=None):
    """models/visual_dialog_model.py:74-120.  `draw(prob)` -> LongTensor[B,1]; defaults to
    torch.multinomial like the reference.  Returns (sequence[B,max_len], per-step filtered logits)."""
    enc_t, enc_v = encoder_forward(sd, enc_cfg, batch["enc_input_ids"], batch["enc_segments"],
                                   batch["enc_attention_mask"], batch["enc_image_features"],
                                   batch["enc_image_spatials"], batch["enc_image_mask"], False)
    enc_h, enc_mask = vl_fusion(sd, enc_t, enc_v, batch["enc_attention_mask"], batch["enc_image_mask"], False)
    dec_ids = batch["dec_input_ids"]
    hist = batch["enc_input_ids"] * (batch["enc_segments"] == 0).long()
    seq, trace = [], []
    for _ in range(max_len):
        y = decoder_hidden(sd, dec_cfg, dec_ids, None, enc_h, enc_mask, False)
        logits = lm_logits(sd, y)[:, -1, :] / temperature
        logits = ngram_block(logits, hist, dec_ids, ngram_blocking_size)
        logits = top_k_top_p_filter(logits, top_k=top_k, top_p=top_p)
        trace.append(logits.clone())
        prob = F.softmax(logits, dim=-1)
        nxt = torch.multinomial(prob, 1) if draw is None else draw(prob)
        dec_ids = torch.cat((dec_ids, nxt), dim=-1)
        seq.append(nxt)
    return pad_after_eos(torch.cat(seq, 1), dec_cfg.get("eos_token_id", 102), dec_cfg.get("pad_token_id", 0)), trace


# ----------------------------------------------------------------------------
# step driver / eval scoring / metrics / schedule
# ----------------------------------------------------------------------------
def candidate_rows(dec_labels_flat):
    """train_gen.py:65-67: rows whose label row is not all zero (as float weights)."""
    return (dec_labels_flat.sum(-1) != 0).float()


def flatten_and_gather(batch, sample_indices):
    """train_gen.py:45-116: flatten [B,10,1,L] -> [10B,L], then row-gather every tensor."""
    out = {}
    for k, v in batch.items():
        if k in ("enc_image_feat", "enc_image_loc", "enc_image_target"):
            flat = v.reshape(-1, v.shape[-2], v.shape[-1])
        elif k in ("enc_hist_len", "enc_next_sentence_labels"):
            flat = v.reshape(-1)
        else:
            flat = v.reshape(-1, v.shape[-1])
        out[k] = flat[sample_indices]
    return out


def answer_scores(logits, dec_input_ids_unmutated):
    """evaluate_gen.py:94-106: sum of log-prob of the left-shifted target ids, pad-masked."""
    lp = F.log_softmax(logits, dim=-1)
    tgt = dec_input_ids_unmutated.new_zeros(dec_input_ids_unmutated.shape)
    tgt[:, :-1] = dec_input_ids_unmutated[:, 1:].clone()
    s = torch.gather(lp, -1, tgt.unsqueeze(-1)).squeeze(-1)
    return (s * (tgt != 0).float()).sum(-1)


def scores_to_ranks(scores):
    """utils/visdial_metrics.py:21-39: 1-based rank of every option (descending score;
    ties resolved by torch.sort's order, as the reference does)."""
    B, Rn, O = scores.shape
    flat = scores.reshape(-1, O)
    order = flat.sort(1, descending=True)[1]
    ranks = torch.empty_like(order)
    ranks.scatter_(1, order, torch.arange(O).expand_as(order))
    return (ranks + 1).view(B, Rn, O)


def sparse_metrics(rank_list):
    """utils/visdial_metrics.py:78-89 from a list/array of ground-truth ranks."""
    r = torch.as_tensor(rank_list).float()
    return {"r@1": (r <= 1).float().mean().item(), "r@5": (r <= 5).float().mean().item(),
            "r@10": (r <= 10).float().mean().item(), "mean": r.mean().item(),
            "mrr": r.reciprocal().mean().item()}


def gt_ranks(scores, gt_option_inds):
    """utils/visdial_metrics.py:51-72."""
    ranks = scores_to_ranks(scores)
    B, Rn, O = ranks.shape
    flat = ranks.view(B * Rn, O)
    return flat[torch.arange(B * Rn), gt_option_inds.view(-1).long()]


def ndcg_batch(scores, relevance):
    """utils/visdial_metrics.py:124-176: per-item NDCG for scores [B,O], relevance [B,O]."""
    ranks = scores_to_ranks(scores.unsqueeze(1)).squeeze(1)
    k = (relevance != 0).sum(-1)
    rankings = torch.sort(ranks, dim=-1)[1]
    best = torch.sort(relevance, dim=-1, descending=True)[1]
    out = []
    for b in range(scores.shape[0]):
        n = int(k[b])
        disc = torch.log2(torch.arange(n).float() + 2)
        dcg = (relevance[b][rankings[b][:n]].float() / disc).sum()
        ideal = (relevance[b][best[b][:n]].float() / disc).sum()
        out.append(dcg / ideal)
    return torch.stack(out)


def warmup_linear_nonzero_lr(step, base_lr, warmup_steps, t_total, min_lr=1e-5):
    """utils/optim_utils.py:19-26."""
    if step < warmup_steps:
        f = float(step) / float(max(1, warmup_steps))
    else:
        f = max(0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))
    return base_lr * f if base_lr * f > min_lr else min_lr


# ----------------------------------------------------------------------------
# helpers for tests / bench
# ----------------------------------------------------------------------------
def live_param_keys(sd):
    """Keys that receive a gradient in enc_dec mode: everything except the
    dead poolers / cls heads / q_dense / sep_embeddings; aliases de-duplicated."""
    dead = ("sep_embeddings", "q_dense", "t_pooler", "v_pooler", ".cls.")
    keys = []
    for k in sd:
        if any(d in k for d in dead):
            continue
        if k.startswith(DEC + "embeddings."):
            continue  # alias of the encoder's embedding module
        if k == LMH + "decoder.bias":
            continue  # alias of lm_head.bias
        keys.append(k)
    return keys


def grads(sd, enc_cfg, dec_cfg, batch, wrt_keys, wrt_feats=True, train=False):
    """loss.backward() through the oracle; returns (outputs, {key: grad}, d loss/d image feats).
    `train`: False (eval), True (host-RNG dropout) or a DropMasks table (injected masks)."""
    sd = {k: v.detach().clone() for k, v in sd.items()}
    # restore aliasing so shared tensors accumulate both contributions
    for k in list(sd):
        if k.startswith(DEC + "embeddings."):
            sd[k] = sd[ENC + "embeddings." + k[len(DEC + "embeddings."):]]
    sd[LMH + "decoder.bias"] = sd[LMH + "bias"]
    for k in wrt_keys:
        sd[k].requires_grad_(True)
    batch = dict(batch)
    feats = batch["enc_image_features"].detach().clone().requires_grad_(wrt_feats)
    batch["enc_image_features"] = feats
    out = model_forward(sd, enc_cfg, dec_cfg, batch, train=train)
    out["loss"].backward()
    g = {k: sd[k].grad for k in wrt_keys}
    return out, g, feats.grad
