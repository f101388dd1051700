"""Container-only pin: the oracle against the LIVE reference at the full bert-base
configuration (config/bert_base_6layer_6conect_{enc,dec}.json).  Skipped wherever
/root/reference is absent (e.g. the GPU box)."""
import json

import pytest
import torch

from oracle import ref_harness as rh
from oracle import vd_oracle as O

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")


def test_full_config_forward_matches_reference():
    e = rh.REFERENCE_ROOT + "/config/bert_base_6layer_6conect_enc.json"
    d = rh.REFERENCE_ROOT + "/config/bert_base_6layer_6conect_dec.json"
    model, params = rh.build_reference_model(e, d, seed=1)
    model.eval()
    enc_cfg, dec_cfg = json.load(open(e)), json.load(open(d))
    g = torch.Generator().manual_seed(5)
    B, T, R, U = 2, 48, 37, 25
    ids = torch.randint(1000, 30000, (B, T), generator=g)
    ids[:, 0] = 101
    ids[1, 30:] = 0
    seg = (torch.arange(T)[None] // 7 % 2).expand(B, T).contiguous()
    att = (ids != 0).float()
    feats = torch.randn(B, R, 2048, generator=g).abs()
    loc = torch.rand(B, R, 5, generator=g)
    im = torch.ones(B, R)
    dec_ids = torch.zeros(B, U, dtype=torch.long)
    dec_ids[:, 0] = 101
    dec_ids[:, 1:6] = torch.randint(1000, 30000, (B, 5), generator=g)
    labels = torch.zeros(B, U, dtype=torch.long)
    labels[:, :5] = dec_ids[:, 1:6]
    labels[:, 5] = 102
    datt = torch.zeros(B, U)
    datt[:, :7] = 1
    with torch.no_grad():
        loss, logits = model(enc_image_features=feats, enc_image_spatials=loc, enc_image_mask=im,
                             enc_input_ids=ids, enc_segments=seg, enc_sep_indices=None, enc_mlm_labels=None,
                             enc_attention_mask=att, dec_input_ids=dec_ids.clone(), dec_attention_mask=datt,
                             dec_labels=labels)
        sd = model.state_dict()
        out = O.model_forward(sd, enc_cfg, dec_cfg, dict(
            enc_input_ids=ids, enc_segments=seg, enc_attention_mask=att, enc_image_features=feats,
            enc_image_spatials=loc, enc_image_mask=im, dec_input_ids=dec_ids.clone(), dec_attention_mask=datt,
            dec_labels=labels))
    assert len(sd) == 861
    assert (out["logits"] - logits).abs().max().item() < 1e-4
    assert abs(out["loss"].item() - loss.item()) < 1e-5
