"""Fixture helpers shared by tests/ and __graft_entry__.smoke(): the tiny-config model of tests/golden/ and its
batch.  Nothing here (or anywhere in this package) imports oracle/ -- the checking itself lives with the callers."""
import json
import os
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def build_tiny_model(precision="fp32", device="cuda:0", mode="vd_train", seed=0, cfg_file="tiny_cfg.json", **extra_params):
    """The tiny-config model of the golden fixtures, weights from tests/golden/tiny_state.npz.  `cfg_file`:
    "tiny_cfg_dropout.json" is the same architecture with a different dropout probability per family (train-mode parity)."""
    from .modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    with open(os.path.join(GOLDEN, cfg_file)) as f:
        cfg = json.load(f)
    d = tempfile.mkdtemp(prefix="gstvd_cfg_")
    with open(os.path.join(d, "enc.json"), "w") as f:
        json.dump(cfg["enc"], f)
    with open(os.path.join(d, "dec.json"), "w") as f:
        json.dump(cfg["dec"], f)
    params = dict(model_enc_config=os.path.join(d, "enc.json"), model_dec_config=os.path.join(d, "dec.json"),
                  gpu_ids=[0], model="enc_dec_a", mode=mode, batch_size=3, device=torch.device(device),
                  amd_precision=precision, amd_seed=seed)
    params.update(extra_params)
    enc, dec = VisualDialogEncoder(params), VisualDialogDecoder(params)
    model = EncoderDecoderModel(params, enc, dec)
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings          # train_gen.py:293
    model.load_state_dict(load_npz("tiny_state.npz"), strict=True)
    return model.to(device), params, cfg


def golden_batch(g, device, dec_key="in::dec_input_ids", with_labels=True):
    b = {k[4:]: v.clone().to(device) for k, v in g.items() if k.startswith("in::")}
    kw = dict(enc_image_features=b["enc_image_features"], enc_image_spatials=b["enc_image_spatials"],
              enc_image_mask=b["enc_image_mask"], enc_input_ids=b["enc_input_ids"], enc_segments=b["enc_segments"],
              enc_attention_mask=b["enc_attention_mask"], dec_input_ids=g[dec_key].clone().to(device),
              dec_attention_mask=b["dec_attention_mask"], dec_labels=b["dec_labels"] if with_labels else None)
    return kw


def dropout_keep_masks(engine):
    """The keep masks the engine's LAST train-mode step applied, one bool tensor per dropout site label (Engine.site_log),
    regenerated on the device by the library's own probe (gstvd_dropout_mask) from the (seed, offset) state that step used and
    the site's number / probability -- i.e. exactly what the fused kernels drew.  Callers (tests, smoke) hand them to the
    oracle's mask hook; "rows" sites index [row, column], "attn" sites [batch, head, query, key] with keys padded to 4."""
    from . import ops
    out = {}
    dev = engine.rng.state.device
    for label, s in engine.site_log.items():
        if s["p"] <= 0:
            continue
        n = 1
        for d in s["shape"]:
            n *= d
        m = ops.dropout_mask(n, s["p"], s["site"], engine.rng, dev).view(s["shape"])
        if s["kind"] == "attn":
            m = m[..., :s["Lk"]]
        out[label] = (m != 0).cpu()
    return out
