"""Host-side mirror of the reference's module API for the enc_dec_a path.

`VisualDialogEncoder`, `VisualDialogDecoder`, `EncoderDecoderModel`, `VLFusion` keep the constructor and
forward signatures, return conventions and the `state_dict()` key layout of

    models/visual_dialog_encoder.py:7-76, models/visual_dialog_decoder.py:18-86,
    models/visual_dialog_model.py:8-135  (+ the parameter tree of models/vilbert_dialog.py)

so that the reference's scripts (train_gen.py:200-202,293; evaluate_gen.py:177-186; generate.py:60-77) can
build, alias (`decoder.decoder.bert.embeddings = encoder.bert_pretrained.bert.embeddings`), load checkpoints
into and call them unchanged.  The sub-modules below are *parameter holders only* (their forward raises):
all arithmetic runs in `engine.Engine` through the HIP C ABI.  There is no CPU execution path.
"""
import torch
from torch import nn

from .config import BertConfig, DecoderConfig


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("gst_visdial_amd parameter holder: compute runs in engine.Engine via the HIP C ABI, "
                           "call EncoderDecoderModel / VisualDialogEncoder / VisualDialogDecoder instead")


class Linear(_Holder):
    def __init__(self, n_in, n_out, bias=True, std=0.02):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in).normal_(0.0, std))
        self.bias = nn.Parameter(torch.zeros(n_out)) if bias else None


class Embedding(_Holder):
    def __init__(self, n, dim, std=0.02):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim).normal_(0.0, std))


class LayerNorm(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class BertEmbeddingsDialog(_Holder):
    """Parameters of models/vilbert_dialog.py:298-322 (the unused sinusoid table is not built)."""

    def __init__(self, cfg):
        super().__init__()
        H, s = cfg.hidden_size, cfg.initializer_range
        self.word_embeddings = Embedding(cfg.vocab_size, H, s)
        self.position_embeddings = Embedding(cfg.max_position_embeddings, H, s)
        self.token_type_embeddings = Embedding(cfg.type_vocab_size, H, s)
        self.token_type_embeddings_extension = Embedding(10, H, s)
        self.sep_embeddings = Embedding(50, H, s)
        self.LayerNorm = LayerNorm(H)


class _SelfAttention(_Holder):
    def __init__(self, hidden, std):
        super().__init__()
        self.query, self.key, self.value = Linear(hidden, hidden, std=std), Linear(hidden, hidden, std=std), Linear(hidden, hidden, std=std)


class _AttnOutput(_Holder):
    def __init__(self, n_in, n_out, std):
        super().__init__()
        self.dense = Linear(n_in, n_out, std=std)
        self.LayerNorm = LayerNorm(n_out)


class _Attention(_Holder):
    def __init__(self, hidden, std):
        super().__init__()
        self.self = _SelfAttention(hidden, std)
        self.output = _AttnOutput(hidden, hidden, std)


class _Intermediate(_Holder):
    def __init__(self, hidden, inter, std):
        super().__init__()
        self.dense = Linear(hidden, inter, std=std)


class _Layer(_Holder):
    """BertLayer / BertImageLayer parameter tree (vilbert_dialog.py:465-476, 592-603)."""

    def __init__(self, hidden, inter, std, cross=False):
        super().__init__()
        self.attention = _Attention(hidden, std)
        if cross:
            self.crossattention = _Attention(hidden, std)
        self.intermediate = _Intermediate(hidden, inter, std)
        self.output = _AttnOutput(inter, hidden, std)


class _BiAttention(_Holder):
    def __init__(self, cfg):
        super().__init__()
        s, Hb = cfg.initializer_range, cfg.bi_hidden_size
        self.query1, self.key1, self.value1 = (Linear(cfg.v_hidden_size, Hb, std=s) for _ in range(3))
        self.query2, self.key2, self.value2 = (Linear(cfg.hidden_size, Hb, std=s) for _ in range(3))


class _BiOutput(_Holder):
    def __init__(self, cfg):
        super().__init__()
        s, Hb = cfg.initializer_range, cfg.bi_hidden_size
        self.dense1 = Linear(Hb, cfg.v_hidden_size, std=s)
        self.LayerNorm1 = LayerNorm(cfg.v_hidden_size)
        self.q_dense1 = Linear(Hb, cfg.v_hidden_size, std=s)      # unused by the reference (vilbert_dialog.py:722)
        self.dense2 = Linear(Hb, cfg.hidden_size, std=s)
        self.LayerNorm2 = LayerNorm(cfg.hidden_size)
        self.q_dense2 = Linear(Hb, cfg.hidden_size, std=s)        # unused by the reference (:729)


class _ConnectionLayer(_Holder):
    """BertConnectionLayer parameter tree (vilbert_dialog.py:746-757)."""

    def __init__(self, cfg):
        super().__init__()
        s = cfg.initializer_range
        self.biattention = _BiAttention(cfg)
        self.biOutput = _BiOutput(cfg)
        self.v_intermediate = _Intermediate(cfg.v_hidden_size, cfg.v_intermediate_size, s)
        self.v_output = _AttnOutput(cfg.v_intermediate_size, cfg.v_hidden_size, s)
        self.t_intermediate = _Intermediate(cfg.hidden_size, cfg.intermediate_size, s)
        self.t_output = _AttnOutput(cfg.intermediate_size, cfg.hidden_size, s)


class _TwoStreamEncoder(_Holder):
    def __init__(self, cfg):
        super().__init__()
        s = cfg.initializer_range
        self.layer = nn.ModuleList([_Layer(cfg.hidden_size, cfg.intermediate_size, s) for _ in range(cfg.num_hidden_layers)])
        self.v_layer = nn.ModuleList([_Layer(cfg.v_hidden_size, cfg.v_intermediate_size, s) for _ in range(cfg.v_num_hidden_layers)])
        self.c_layer = nn.ModuleList([_ConnectionLayer(cfg) for _ in range(len(cfg.v_biattention_id))])


class _ImageEmbeddings(_Holder):
    def __init__(self, cfg):
        super().__init__()
        s = cfg.initializer_range
        self.image_embeddings = Linear(cfg.v_feature_size, cfg.v_hidden_size, std=s)
        self.image_location_embeddings = Linear(5, cfg.v_hidden_size, std=s)
        self.LayerNorm = LayerNorm(cfg.v_hidden_size)


class _Pooler(_Holder):
    def __init__(self, n_in, n_out, std):
        super().__init__()
        self.dense = Linear(n_in, n_out, std=std)


class _BertModel(_Holder):
    """BertModel parameter tree (vilbert_dialog.py:1310-1323)."""

    def __init__(self, cfg):
        super().__init__()
        self.embeddings = BertEmbeddingsDialog(cfg)
        self.v_embeddings = _ImageEmbeddings(cfg)
        self.encoder = _TwoStreamEncoder(cfg)
        self.t_pooler = _Pooler(cfg.hidden_size, cfg.bi_hidden_size, cfg.initializer_range)       # dead in enc_dec
        self.v_pooler = _Pooler(cfg.v_hidden_size, cfg.bi_hidden_size, cfg.initializer_range)     # dead in enc_dec


class _HeadTransform(_Holder):
    def __init__(self, hidden, std):
        super().__init__()
        self.dense = Linear(hidden, hidden, std=std)
        self.LayerNorm = LayerNorm(hidden)


class _LMPredictionHead(_Holder):
    def __init__(self, cfg, tied_weight):
        super().__init__()
        self.transform = _HeadTransform(cfg.hidden_size, cfg.initializer_range)
        self.decoder = Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
        self.decoder.weight = tied_weight                      # tied to the word embedding (vilbert_dialog.py:991)
        self.bias = nn.Parameter(torch.zeros(cfg.vocab_size))


class _ImagePredictionHead(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.transform = _HeadTransform(cfg.v_hidden_size, cfg.initializer_range)
        self.decoder = Linear(cfg.v_hidden_size, cfg.v_target_size, std=cfg.initializer_range)


class _PreTrainingHeads(_Holder):
    """cls.* of BertForMultiModalPreTraining (vilbert_dialog.py:1017-1024): dead compute in enc_dec mode
    (outputs discarded at :1485-1487); kept only so checkpoints load with strict key matching."""

    def __init__(self, cfg, tied_weight):
        super().__init__()
        self.predictions = _LMPredictionHead(cfg, tied_weight)
        self.bi_seq_relationship = Linear(cfg.bi_hidden_size, 2, std=cfg.initializer_range)
        self.imagePredictions = _ImagePredictionHead(cfg)


class BertForMultiModalPreTraining(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.bert = _BertModel(cfg)
        self.cls = _PreTrainingHeads(cfg, self.bert.embeddings.word_embeddings.weight)


class _GenerationLMHead(_Holder):
    """BertGenerationOnlyLMHead (visual_dialog_decoder.py:326-343): `decoder.weight` starts tied to the decoder's
    own word embedding and stays a separate parameter once the embedding module is replaced by the encoder's;
    `bias` and `decoder.bias` are one parameter under two names."""

    def __init__(self, tied_weight):
        super().__init__()
        V, H = tied_weight.shape
        self.decoder = Linear(H, V)
        self.decoder.weight = tied_weight
        self.bias = nn.Parameter(torch.zeros(V))
        self.decoder.bias = self.bias


class _GenerationStack(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg.hidden_size, cfg.intermediate_size, cfg.initializer_range, cross=True)
                                    for _ in range(cfg.num_hidden_layers)])


class BertGenerationEncoder(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.embeddings = BertEmbeddingsDialog(cfg)
        self.encoder = _GenerationStack(cfg)


class BertForSequenceGeneration(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.bert = BertGenerationEncoder(cfg)
        self.lm_head = _GenerationLMHead(self.bert.embeddings.word_embeddings.weight)

    def _reorder_cache(self, past, beam_idx):   # exists in the reference (visual_dialog_decoder.py:177-181), never called
        return tuple(tuple(s.index_select(0, beam_idx) for s in layer) for layer in past)


class VLFusion(_Holder):
    """models/visual_dialog_model.py:123-135."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.fc_l = Linear(config.hidden_size, config.hidden_size)
        self.fc_v = Linear(config.v_hidden_size, config.hidden_size)


class DecoderOutput(object):
    """The fields of transformers' Seq2SeqLMOutput the reference reads (visual_dialog_model.py:72)."""

    def __init__(self, loss, logits):
        self.loss, self.logits = loss, logits
        self.past_key_values = self.decoder_hidden_states = self.decoder_attentions = self.cross_attentions = None


def _check_master_current(engine, what):
    """state_dict() of the model OR of one of its sub-modules (train_gen.py:346-357 saves `model.module.encoder` /
    `.decoder` state dicts separately in some forks): refused while a sharded optimizer holds stale fp32 masters here."""
    pipe = getattr(engine, "pipe", None) if engine is not None else None
    if pipe is not None and hasattr(pipe, "check_master_current"):
        pipe.check_master_current(what)


class VisualDialogEncoder(nn.Module):
    """models/visual_dialog_encoder.py:7-76.  `params` is held by reference and re-read on every call."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.config = BertConfig.from_json_file(params["model_enc_config"])
        self.config.__dict__["cur_device"] = params["gpu_ids"][0]
        self.config.__dict__["model_arch"] = params["model"]
        self.config.__dict__["mode"] = params["mode"]
        self.config.validate()
        self.model_arch = params["model"]
        if "enc_dec" not in self.model_arch:
            raise NotImplementedError("gst_visdial_amd implements the enc_dec_* generative path only (model=%r)" % self.model_arch)
        # the reference calls from_pretrained('bert-base-uncased') (network); here weights come from
        # load_state_dict / a checkpoint, with BERT-style N(0, 0.02) init as the starting point
        self.bert_pretrained = BertForMultiModalPreTraining(self.config)
        self._engine_owner = None

    def state_dict(self, *args, **kwargs):
        if not kwargs.get("prefix") and not (len(args) > 1 and args[1]):      # (a parent's state_dict() has checked already)
            _check_master_current(getattr(self, "_standalone_engine", None), "model.encoder.state_dict()")
        return super().state_dict(*args, **kwargs)

    def forward(self, input_ids, image_feat, image_loc, sep_indices=None, token_type_ids=None, attention_mask=None,
                masked_lm_labels=None, next_sentence_label=None, image_attention_mask=None, image_label=None,
                image_target=None):
        from .engine import standalone_encoder_forward
        enc_t, enc_v = standalone_encoder_forward(self, input_ids, image_feat, image_loc, token_type_ids, attention_mask,
                                                  image_attention_mask)
        return (None, None, None, None, None, enc_t, enc_v)


class VisualDialogDecoder(nn.Module):
    """models/visual_dialog_decoder.py:18-86."""

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.config = DecoderConfig.from_json_file(params["model_dec_config"])
        self.config.__dict__["cur_device"] = params["gpu_ids"][0]
        self.config.validate()
        self.decoder = BertForSequenceGeneration(self.config)

    def state_dict(self, *args, **kwargs):
        if not kwargs.get("prefix") and not (len(args) > 1 and args[1]):
            _check_master_current(getattr(self, "_standalone_engine", None), "model.decoder.state_dict()")
        return super().state_dict(*args, **kwargs)

    def _reorder_cache(self, past, beam_idx):
        return self.decoder._reorder_cache(past, beam_idx)

    def forward(self, decoder_input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, use_cache=False, output_attentions=False, output_hidden_states=False, return_dict=True,
                loss_reduction=True):
        from .engine import standalone_decoder_forward
        loss, logits = standalone_decoder_forward(self, decoder_input_ids, attention_mask, encoder_hidden_states,
                                                  encoder_attention_mask, labels, loss_reduction)
        return DecoderOutput(loss, logits)


class EncoderDecoderModel(nn.Module):
    """models/visual_dialog_model.py:8-120: encoder -> VLFusion -> decoder (train/eval) or the 18-step sampling
    decode.  One `torch.autograd.Function` wraps the whole step, so `loss.backward()` runs the hand-written
    backward and fills `.grad` of every live parameter (and of `enc_image_features` when it requires grad)."""

    def __init__(self, params, encoder, decoder):
        super().__init__()
        self.params = params
        self.encoder = encoder
        self.decoder = decoder
        self.vlfusion = VLFusion(encoder.config)
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self)
            object.__setattr__(self.encoder, "_standalone_engine", self._engine)
            object.__setattr__(self.decoder, "_standalone_engine", self._engine)
        return self._engine

    def state_dict(self, *args, **kwargs):
        _check_master_current(self._engine, "model.state_dict()")
        return super().state_dict(*args, **kwargs)

    def _replicate_for_data_parallel(self):
        """nn.DataParallel(model, [0, 1, 2, 3]) (train_gen.py:295, README.md:89) replicates the module tree per device and runs the
        replicas in threads; the replicas would all drive ONE engine (one flat parameter buffer, one arena, one tape).  Refuse
        loudly instead: this framework scales as one process per GPU.  (`nn.DataParallel(model, [0])` never replicates.)"""
        from ._lib import GstvdError
        raise GstvdError(
            "gst_visdial_amd.EncoderDecoderModel cannot be replicated by nn.DataParallel over several device ids: it runs one "
            "process per GPU.  Keep nn.DataParallel(model, [local_gpu]) for the reference's `.module` access and launch N ranks "
            "(torchrun --nproc-per-node N train script) with gst_visdial_amd.pipeline.BackwardPipeline doing the gradient "
            "all-reduce over RCCL (INTEGRATION.md, 'Multi-GPU').")

    def score_candidates(self, enc_image_features, enc_image_spatials, enc_image_mask, enc_input_ids, enc_segments,
                         enc_attention_mask, dec_input_ids, dec_attention_mask, num_options):
        """Generative ranking scores of evaluate_gen.py:45-106 with one encoder pass per dialog round (see
        Engine.score_candidates).  Encoder tensors: one row per round; decoder tensors: num_options rows per round."""
        return self.engine.score_candidates(enc_image_features, enc_image_spatials, enc_image_mask, enc_input_ids,
                                            enc_segments, enc_attention_mask, dec_input_ids, dec_attention_mask, num_options)

    def forward(self, enc_image_features=None, enc_image_spatials=None, enc_image_mask=None, enc_image_target=None,
                enc_image_label=None, enc_next_sentence_labels=None, enc_input_ids=None, enc_segments=None,
                enc_sep_indices=None, enc_mlm_labels=None, enc_attention_mask=None, dec_input_ids=None,
                dec_attention_mask=None, dec_labels=None, loss_reduction=True, **decoding_kwargs):
        mode = self.params["mode"]
        if "train" in mode or "eval" in mode:
            return self.engine.step(enc_image_features, enc_image_spatials, enc_image_mask, enc_input_ids, enc_segments,
                                    enc_attention_mask, dec_input_ids, dec_attention_mask, dec_labels, loss_reduction)
        return self.engine.sample(enc_image_features, enc_image_spatials, enc_image_mask, enc_input_ids, enc_segments,
                                  enc_attention_mask, dec_input_ids, **decoding_kwargs)
