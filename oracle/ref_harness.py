"""Harness that imports the REAL reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  This file runs only in the build container (the
reference tree does not exist on the GPU box) and is used by
`oracle/make_golden.py` to emit the golden vectors under `tests/golden/` and
by `tests/test_oracle_vs_reference.py` (skipped when /root/reference is
absent) to pin `oracle/vd_oracle.py` against the reference itself.

Nothing here is copied from the reference: it only *imports* it, after
installing the shims listed in SURVEY.md Appendix A:

  1. stub modules for the two legacy packages the reference imports but does
     not use on this path (models/vilbert_dialog.py:34,37);
  2. ignore `.to(cuda)` for the unused sinusoid table (vilbert_dialog.py:312);
  3. construct from config instead of downloading bert-base-uncased
     (models/visual_dialog_encoder.py:17, models/visual_dialog_decoder.py:24);
  4. restore transformers-4.16.2 semantics of the three mask helpers the
     reference's own `BertGenerationEncoder` calls
     (models/visual_dialog_decoder.py:274,285,294) -- the installed
     transformers is 5.x.
"""
import json
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


_installed = {}


def _install_shims():
    if _installed:
        return _installed
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    pt = types.ModuleType("pytorch_transformers")
    mb = types.ModuleType("pytorch_transformers.modeling_bert")
    mb.BertEmbeddings = object
    pt.modeling_bert = mb
    sys.modules.setdefault("pytorch_transformers", pt)
    sys.modules.setdefault("pytorch_transformers.modeling_bert", mb)
    pp = types.ModuleType("pytorch_pretrained_bert")
    fu = types.ModuleType("pytorch_pretrained_bert.file_utils")
    fu.cached_path = lambda p, cache_dir=None: p
    pp.file_utils = fu
    sys.modules.setdefault("pytorch_pretrained_bert", pp)
    sys.modules.setdefault("pytorch_pretrained_bert.file_utils", fu)

    _to = torch.Tensor.to

    def _cpu_only_to(self, *a, **k):
        if a and isinstance(a[0], torch.device) and a[0].type == "cuda" and not torch.cuda.is_available():
            return self
        return _to(self, *a, **k)

    torch.Tensor.to = _cpu_only_to

    import models.visual_dialog_encoder as E
    import models.visual_dialog_decoder as D
    from models.visual_dialog_model import EncoderDecoderModel

    E.BertForMultiModalPreTraining.from_pretrained = classmethod(
        lambda cls, name, config, *a, **k: cls(config))
    D.BertForSequenceGeneration.from_pretrained = classmethod(
        lambda cls, name, config=None, *a, **k: cls(config))

    def _ext(self, m, shape, device=None):
        B, T = shape
        if m.dim() == 2 and self.config.is_decoder:
            i = torch.arange(T)
            causal = (i[None, None, :].repeat(B, T, 1) <= i[None, :, None]).to(m.dtype)
            m = causal[:, None] * m[:, None, None, :]
        else:
            m = m[:, None, None, :] if m.dim() == 2 else m[:, None]
        return (1.0 - m.float()) * -10000.0

    D.BertGenerationEncoder.get_extended_attention_mask = _ext
    D.BertGenerationEncoder.invert_attention_mask = lambda self, m: (
        1.0 - (m[:, None, None, :] if m.dim() == 2 else m[:, None]).float()) * -1e9
    D.BertGenerationEncoder.get_head_mask = lambda self, hm, n, *a, **k: [None] * n

    _installed.update(E=E, D=D, EncoderDecoderModel=EncoderDecoderModel)
    return _installed


def build_reference_model(enc_cfg_path, dec_cfg_path, mode="vd_train", seed=0, batch_size=1):
    """Construct the reference EncoderDecoderModel on CPU with seeded init."""
    mods = _install_shims()
    E, D = mods["E"], mods["D"]
    params = dict(model_enc_config=enc_cfg_path, model_dec_config=dec_cfg_path,
                  gpu_ids=[0], model="enc_dec_a", mode=mode, batch_size=batch_size,
                  device=torch.device("cpu"))
    torch.manual_seed(seed)
    enc = E.VisualDialogEncoder(params)
    dec = D.VisualDialogDecoder(params)
    dec.decoder.apply(dec.decoder._init_weights)
    model = mods["EncoderDecoderModel"](params, enc, dec)
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings  # train_gen.py:293
    return model, params


def reference_utils():
    _install_shims()
    import utils.decoding_utils as du
    import utils.visdial_metrics as vm
    import utils.optim_utils as ou
    return du, vm, ou


TINY_ENC_CFG = {
    "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu", "hidden_dropout_prob": 0.3,
    "hidden_size": 64, "initializer_range": 0.02, "intermediate_size": 128,
    "max_position_embeddings": 48, "model_type": "bert-generation",
    "num_attention_heads": 2, "num_hidden_layers": 4, "type_vocab_size": 2, "vocab_size": 320,
    "v_feature_size": 40, "v_target_size": 11, "v_hidden_size": 96, "v_num_hidden_layers": 2,
    "v_num_attention_heads": 3, "v_intermediate_size": 80, "bi_hidden_size": 128,
    "bi_num_attention_heads": 4, "bi_intermediate_size": 128, "bi_attention_type": 1,
    "v_attention_probs_dropout_prob": 0.1, "v_hidden_act": "gelu", "v_hidden_dropout_prob": 0.3,
    "v_initializer_range": 0.02, "v_biattention_id": [0, 1], "t_biattention_id": [2, 3],
    "pooling_method": "mul",
}
TINY_DEC_CFG = {
    "attention_probs_dropout_prob": 0.1, "hidden_act": "gelu", "hidden_dropout_prob": 0.3,
    "hidden_size": 64, "initializer_range": 0.02, "intermediate_size": 128,
    "max_position_embeddings": 48, "model_type": "bert-generation",
    "num_attention_heads": 2, "num_hidden_layers": 2, "type_vocab_size": 2, "vocab_size": 320,
    "v_feature_size": 40, "v_target_size": 11, "v_hidden_size": 96, "v_num_hidden_layers": 2,
    "v_num_attention_heads": 3, "v_intermediate_size": 80,
    "add_cross_attention": True, "is_decoder": True, "layer_norm_eps": 1e-12,
    "bos_token_id": 101, "eos_token_id": 102, "use_cache": False,
    "decoder_start_token_id": 101, "pad_token_id": 0,
}


def write_tiny_configs(dirpath):
    os.makedirs(dirpath, exist_ok=True)
    e = os.path.join(dirpath, "tiny_enc.json")
    d = os.path.join(dirpath, "tiny_dec.json")
    with open(e, "w") as f:
        json.dump(TINY_ENC_CFG, f, indent=1)
    with open(d, "w") as f:
        json.dump(TINY_DEC_CFG, f, indent=1)
    return e, d
