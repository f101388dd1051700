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


def build_tiny_model(precision="fp32", device="cuda:0", mode="vd_train", seed=0, cfg_file="tiny_cfg.json", state_file="tiny_state.npz",
                     **extra_params):
    """The tiny-config model of the golden fixtures, weights from tests/golden/tiny_state.npz.  `cfg_file`:
    "tiny_cfg_dropout.json" is the same architecture with a different dropout probability per family (train-mode parity);
    `state_file`: "tiny_state_trained.npz" is the TRAINED tiny checkpoint of oracle/make_golden_r4.py (peaked answer distributions)."""
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
    model.load_state_dict(load_npz(state_file), strict=True)
    return model.to(device), params, cfg


def evalset100_batches(ev, dialogs_per_batch=2):
    """tests/golden/tiny_evalset100.npz -> batches in the eval dataloader's layout: the fixture stores a
    round's context once; the loader repeats it for each of the round's 100 options (dataloader_visdial_gen.py:379-388)."""
    b = {k[4:]: v for k, v in ev.items() if k.startswith("in::")}
    G = b["dec_input_ids"].shape[2]
    ids = b["enc_input_ids"].long()
    full = dict(enc_input_ids=ids[:, :, None].expand(-1, -1, G, -1).contiguous(),
                enc_segments=b["enc_segments"].long()[:, :, None].expand(-1, -1, G, -1).contiguous(),
                dec_input_ids=b["dec_input_ids"].long(), dec_att_mask=b["dec_att_mask"], enc_image_feat=b["enc_image_feat"],
                enc_image_loc=b["enc_image_loc"], enc_image_mask=b["enc_image_mask"], gt_option_inds=b["gt_option_inds"],
                gt_relevance=b["gt_relevance"], round_id=b["round_id"])
    full["enc_att_mask"] = (full["enc_input_ids"] != 0).float()
    n = ids.shape[0]
    return [{k: v[s:s + dialogs_per_batch].clone() for k, v in full.items()} for s in range(0, n, dialogs_per_batch)]


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
