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


def build_tiny_model(precision="fp32", device="cuda:0", mode="vd_train", seed=0):
    """The tiny-config model of the golden fixtures, weights from tests/golden/tiny_state.npz."""
    from .modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    with open(os.path.join(GOLDEN, "tiny_cfg.json")) as f:
        cfg = json.load(f)
    d = tempfile.mkdtemp(prefix="gstvd_cfg_")
    with open(os.path.join(d, "enc.json"), "w") as f:
        json.dump(cfg["enc"], f)
    with open(os.path.join(d, "dec.json"), "w") as f:
        json.dump(cfg["dec"], f)
    params = dict(model_enc_config=os.path.join(d, "enc.json"), model_dec_config=os.path.join(d, "dec.json"),
                  gpu_ids=[0], model="enc_dec_a", mode=mode, batch_size=3, device=torch.device(device),
                  amd_precision=precision, amd_seed=seed)
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
