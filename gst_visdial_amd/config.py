"""Model configuration objects of the hot path.

`BertConfig` mirrors what models/vilbert_dialog.py:131-274 exposes to callers (attribute access,
`from_json_file`, `from_dict`, defaults for keys the shipped JSON omits) and `DecoderConfig` what
the reference takes from transformers' `BertGenerationConfig` (visual_dialog_decoder.py:22):
`vocab_size`, `eos_token_id`, `pad_token_id`, `layer_norm_eps`, ... -- without depending on
`transformers`.
"""
import copy
import json

_ENC_DEFAULTS = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
    type_vocab_size=2, initializer_range=0.02, v_feature_size=2048, v_target_size=1601, v_hidden_size=768,
    v_num_hidden_layers=3, v_num_attention_heads=12, v_intermediate_size=3072, bi_hidden_size=1024,
    bi_num_attention_heads=16, v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu", v_hidden_dropout_prob=0.1,
    v_initializer_range=0.2, v_biattention_id=[0, 1], t_biattention_id=[10, 11], predict_feature=False,
    fast_mode=False, fixed_v_layer=0, fixed_t_layer=0, in_batch_pairs=False, fusion_method="mul", intra_gate=False,
    with_coattention=True,
)

_DEC_DEFAULTS = dict(
    vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
    type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0, bos_token_id=2, eos_token_id=1,
    is_decoder=False, add_cross_attention=False, use_cache=True,
)


class _Cfg(object):
    _defaults = {}

    def __init__(self, **kw):
        self.__dict__.update(copy.deepcopy(self._defaults))
        self.__dict__.update(kw)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def __repr__(self):
        return self.to_json_string()


class BertConfig(_Cfg):
    """Encoder (two-stream ViLBERT) configuration; config/bert_base_6layer_6conect_enc.json."""
    _defaults = _ENC_DEFAULTS

    def validate(self):
        assert len(self.v_biattention_id) == len(self.t_biattention_id)
        assert max(self.v_biattention_id) < self.v_num_hidden_layers
        assert max(self.t_biattention_id) < self.num_hidden_layers
        for k in ("fast_mode", "in_batch_pairs", "fixed_v_layer", "fixed_t_layer"):
            if getattr(self, k):
                raise NotImplementedError("config.%s is not on the enc_dec_a path (reference default is off)" % k)
        if not self.with_coattention:
            raise NotImplementedError("with_coattention=False is not on the enc_dec_a path")
        if self.hidden_act != "gelu" or self.v_hidden_act != "gelu":
            raise NotImplementedError("only the erf GELU of the shipped configs is implemented")


class DecoderConfig(_Cfg):
    """BERT-generation decoder configuration; config/bert_base_6layer_6conect_dec.json."""
    _defaults = _DEC_DEFAULTS

    def validate(self):
        if not (self.is_decoder and self.add_cross_attention):
            raise NotImplementedError("the decoder must be configured with is_decoder and add_cross_attention")
        if self.hidden_act != "gelu":
            raise NotImplementedError("only the erf GELU of the shipped configs is implemented")


def encoder_schedule(cfg):
    """Sub-layer order of BertEncoder.forward (models/vilbert_dialog.py:831-905): ('t'|'v'|'c', index)."""
    order, vs, ts = [], 0, 0
    for c, (ve, te) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        order += [("v", i) for i in range(vs, ve)]
        order += [("t", i) for i in range(ts, te)]
        order.append(("c", c))
        vs, ts = ve, te
    order += [("v", i) for i in range(vs, cfg.v_num_hidden_layers)]
    order += [("t", i) for i in range(ts, cfg.num_hidden_layers)]
    return order


def bert_base_enc_config():
    """The values of the reference's config/bert_base_6layer_6conect_enc.json (12 text / 6 vision / 6 connection layers)."""
    return dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.3, hidden_size=768,
                initializer_range=0.02, intermediate_size=3072, max_position_embeddings=512, model_type="bert-generation",
                num_attention_heads=12, num_hidden_layers=12, type_vocab_size=2, vocab_size=30522, v_feature_size=2048,
                v_target_size=1601, v_hidden_size=1024, v_num_hidden_layers=6, v_num_attention_heads=8,
                v_intermediate_size=1024, bi_hidden_size=1024, bi_num_attention_heads=8, bi_intermediate_size=1024,
                bi_attention_type=1, v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu", v_hidden_dropout_prob=0.3,
                v_initializer_range=0.02, v_biattention_id=[0, 1, 2, 3, 4, 5], t_biattention_id=[6, 7, 8, 9, 10, 11],
                pooling_method="mul")


def bert_base_dec_config():
    """The values of the reference's config/bert_base_6layer_6conect_dec.json (12-layer decoder with cross-attention)."""
    return dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.3, hidden_size=768,
                initializer_range=0.02, intermediate_size=3072, max_position_embeddings=512, model_type="bert-generation",
                num_attention_heads=12, num_hidden_layers=12, type_vocab_size=2, vocab_size=30522, v_feature_size=2048,
                v_target_size=1601, v_hidden_size=1024, v_num_hidden_layers=6, v_num_attention_heads=8,
                v_intermediate_size=1024, add_cross_attention=True, is_decoder=True, layer_norm_eps=1e-12, bos_token_id=101,
                eos_token_id=102, use_cache=False, decoder_start_token_id=101, pad_token_id=0)
