"""Round-3 golden vectors: the REFERENCE's train step in TRAIN mode (dropout ON) under injected dropout masks.

Run only in the build container:  python oracle/make_golden_r3.py     (imports /root/reference through oracle/ref_harness.py)

Why: the timed configuration trains with dropout (train_gen.py:300,324); every earlier fixture ran `model.eval()`.  A
Bernoulli draw cannot be compared across implementations, a MASK can: here `torch.nn.functional.dropout` is replaced, for
the duration of the reference's forward, by  x * mask / (1 - p)  with
  * `p`     the probability the reference's own call site passes (its nn.Dropout module's `.p`, or the functional call of
            transformers' eager attention) -- so the fixture records WHICH probability reaches WHICH site,
  * `mask`  a seeded 0/1 tensor stored in the fixture under the oracle's label of the site (oracle/vd_oracle.py, DropMasks).
The model config gives every dropout family its own probability (text 0.3 / 0.1, vision 0.2 / 0.15, decoder 0.25 / 0.05), so
a site that takes the wrong family's value changes the outputs.  Note what the fixture pins for the decoder's embedding
dropout: the embedding MODULE is the encoder's (train_gen.py:293), so its nn.Dropout carries the ENCODER's
hidden_dropout_prob also when the decoder calls it.

Written:
  tiny_cfg_dropout.json   {"enc": {...}, "dec": {...}}  the tiny config with per-family dropout probabilities
  tiny_train_dropout.npz  mask::<label> (uint8), p::<label> (the probability the reference passed), site order, loss, logits,
                          the fused encoder states, the 29 golden-key gradients, d loss / d image features
                          (weights: tests/golden/tiny_state.npz, inputs: tests/golden/tiny_train.npz `in::*`)
"""
import json
import os
import re
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh      # noqa: E402
from oracle import make_golden as mg      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

ENC_P = dict(hidden_dropout_prob=0.3, attention_probs_dropout_prob=0.1, v_hidden_dropout_prob=0.2,
             v_attention_probs_dropout_prob=0.15)
DEC_P = dict(hidden_dropout_prob=0.25, attention_probs_dropout_prob=0.05)

_BERT = "encoder.bert_pretrained.bert."
_DECL = "decoder.decoder.bert.encoder.layer."


def site_label(module_path, call_index):
    """Reference module path of a dropout call (+ how often that module has been called before) -> the oracle's site label;
    None for the dead heads (encoder.bert_pretrained.cls.*, vilbert_dialog.py:1482: outputs discarded in enc_dec)."""
    p = module_path
    if p == _BERT + "embeddings.dropout":
        return ("emb.enc", "emb.dec")[call_index]          # shared module: encoder call first, decoder call second
    if p == _BERT + "v_embeddings.dropout":
        return "vemb"
    if p == "vlfusion.dropout":
        return "vlf"
    m = re.match(re.escape(_BERT) + r"encoder\.(layer|v_layer)\.(\d+)\.(attention\.self|attention\.output|output)\.dropout$", p)
    if m:
        return "%s%s.%s" % ("t" if m.group(1) == "layer" else "v", m.group(2),
                            {"attention.self": "attn", "attention.output": "ln1", "output": "ln2"}[m.group(3)])
    m = re.match(re.escape(_BERT) + r"encoder\.c_layer\.(\d+)\.(biattention\.dropout1|biattention\.dropout2|biOutput\.dropout1|"
                 r"biOutput\.dropout2|v_output\.dropout|t_output\.dropout)$", p)
    if m:
        return "c%s.%s" % (m.group(1), {"biattention.dropout1": "attn1", "biattention.dropout2": "attn2", "biOutput.dropout1": "ln1",
                                        "biOutput.dropout2": "ln2", "v_output.dropout": "vln", "t_output.dropout": "tln"}[m.group(2)])
    m = re.match(re.escape(_DECL) + r"(\d+)\.(attention\.self|attention\.output\.dropout|crossattention\.self|"
                 r"crossattention\.output\.dropout|output\.dropout)$", p)
    if m:
        return "d%s.%s" % (m.group(1), {"attention.self": "attn", "attention.output.dropout": "ln1", "crossattention.self": "xattn",
                                        "crossattention.output.dropout": "ln2", "output.dropout": "ln3"}[m.group(2)])
    if p.startswith("encoder.bert_pretrained.cls"):
        return None
    raise KeyError("unmapped dropout site in the reference: %s" % p)


def seeded_mask(label, shape, p):
    g = torch.Generator().manual_seed(zlib.crc32(label.encode()))
    return (torch.rand(shape, generator=g) >= p).to(torch.uint8)


def main():
    enc_cfg = dict(rh.TINY_ENC_CFG, **ENC_P)
    dec_cfg = dict(rh.TINY_DEC_CFG, **DEC_P)
    cfg_dir = os.path.join(OUT, "_cfg_r3")
    os.makedirs(cfg_dir, exist_ok=True)
    e_path, d_path = os.path.join(cfg_dir, "enc.json"), os.path.join(cfg_dir, "dec.json")
    with open(e_path, "w") as f:
        json.dump(enc_cfg, f)
    with open(d_path, "w") as f:
        json.dump(dec_cfg, f)
    model, params = rh.build_reference_model(e_path, d_path, mode="vd_train", seed=0)
    with np.load(os.path.join(OUT, "tiny_state.npz")) as z:
        model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    with np.load(os.path.join(OUT, "tiny_train.npz")) as z:
        inp = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in::")}
    model.train()                                         # train_gen.py:300

    # which module is executing when F.dropout is called
    stack, names, ncalls = [], {}, {}
    for n, m in model.named_modules():
        names[id(m)] = n
        m.register_forward_pre_hook(lambda mod, a: stack.append(names[id(mod)]))
        m.register_forward_hook(lambda mod, a, o: (stack.pop(), None)[1])
    masks, probs, order = {}, {}, []
    import torch.nn.functional as F
    orig = F.dropout

    def injected(x, p=0.5, training=True, inplace=False):
        assert training and not inplace, (stack[-1], training, inplace)
        path = stack[-1]
        k = ncalls.get(path, 0)
        ncalls[path] = k + 1
        label = site_label(path, k)
        if label is None:
            return orig(x, p, training, False)            # dead head: any draw
        if label == "vlf":                                # one call over cat(vision rows, text rows): two oracle sites
            R = inp["enc_image_features"].shape[1]
            mv, ml = seeded_mask("vlf.v", x[:, :R].shape, p), seeded_mask("vlf.l", x[:, R:].shape, p)
            masks["vlf.v"], masks["vlf.l"] = mv, ml
            probs["vlf.v"] = probs["vlf.l"] = p
            order.extend(["vlf.v", "vlf.l"])
            mk = torch.cat((mv, ml), dim=1)
        else:
            assert label not in masks, label
            mk = masks[label] = seeded_mask(label, x.shape, p)
            probs[label] = p
            order.append(label)
        return x * (mk.to(x.dtype) * (1.0 / (1.0 - p)))

    F.dropout = injected
    try:
        feats = inp["enc_image_features"].clone().requires_grad_(True)
        captured = {}
        model.vlfusion.register_forward_hook(lambda m, i, o: captured.update(fused=o[0]))
        model.zero_grad()
        loss, logits = mg.call_model(model, dict(inp, enc_image_features=feats), inp["dec_input_ids"].clone(), inp["dec_labels"])
        loss.backward()
    finally:
        F.dropout = orig
    named = dict(model.named_parameters())
    out = dict(loss=loss, logits=logits, d_feats=feats.grad, enc_hidden=captured["fused"])
    for k in mg.GRAD_KEYS:
        out["grad::" + k] = named[k].grad.clone()
    for lab in order:
        out["mask::" + lab] = masks[lab]
        out["p::" + lab] = np.float64(probs[lab])
    np.savez_compressed(os.path.join(OUT, "tiny_train_dropout.npz"), **mg.npy(out))
    with open(os.path.join(OUT, "tiny_cfg_dropout.json"), "w") as f:
        json.dump({"enc": enc_cfg, "dec": dec_cfg, "site_order": order}, f, indent=1, sort_keys=True)
    import shutil
    shutil.rmtree(cfg_dir)
    print("%d dropout sites; loss %.6f" % (len(order), float(loss)))
    print(" ".join("%s=%.2f" % (l, probs[l]) for l in order))


if __name__ == "__main__":
    main()
