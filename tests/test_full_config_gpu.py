"""Parity at the reference's FULL model size (BASELINE.json configs[0]: 2 dialog rounds, seq_len 128, 37x2048 region
features, bert_base_6layer_6conect encoder + 12-layer decoder, 388 M parameters) against the CPU oracle, which is itself
pinned to the reference at this config (tests/test_oracle_vs_reference.py).  fp32 mode: logits within 1e-4 (north_star
tolerance); bf16 mode: loss / logits within bf16 noise.  A handful of gradients across the whole depth of the network
are checked too, and the configs[1] shape (16 rows, seq_len 256) is covered by size-independent properties."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _oracle():
    from oracle import vd_oracle as O
    return O


@pytest.fixture(scope="module")
def full_fp32():
    import bench
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    torch.manual_seed(0)
    model, params = bench.build_model(torch.device(DEV), "fp32", seed=7)
    model.eval()
    V = model.decoder.config.vocab_size
    batch = bench.synthetic_rows(2, 128, 37, 25, 2048, V, 4321, DEV)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cpu_batch = {k: v.cpu() for k, v in batch.items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    O = _oracle()
    keys = ["encoder.bert_pretrained.bert.embeddings.word_embeddings.weight",
            "encoder.bert_pretrained.bert.v_embeddings.image_embeddings.weight",
            "encoder.bert_pretrained.bert.encoder.layer.0.attention.self.query.weight",
            "encoder.bert_pretrained.bert.encoder.v_layer.3.intermediate.dense.weight",
            "encoder.bert_pretrained.bert.encoder.c_layer.2.biattention.key2.weight",
            "encoder.bert_pretrained.bert.encoder.c_layer.5.biOutput.dense1.bias",
            "encoder.bert_pretrained.bert.encoder.layer.11.output.LayerNorm.weight",
            "vlfusion.fc_v.weight",
            "decoder.decoder.bert.encoder.layer.0.crossattention.self.key.weight",
            "decoder.decoder.bert.encoder.layer.11.output.dense.weight",
            "decoder.decoder.lm_head.decoder.weight"]
    keys = [k for k in keys if k in sd]
    assert len(keys) >= 9, keys
    out, g, dfe = O.grads(sd, bert_base_enc_config(), bert_base_dec_config(), cpu_batch, keys, wrt_feats=True)
    ref = dict(logits=out["logits"].detach(), loss=out["loss"].detach(), grads=g, dfeats=dfe)
    return model, batch, sd, cpu_batch, keys, ref


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def test_full_model_fp32_logits_and_loss_match_oracle(full_fp32):
    model, batch, sd, cpu_batch, keys, ref = full_fp32
    feats = batch["enc_image_features"].clone().requires_grad_(True)
    kw = dict(batch, enc_image_features=feats)
    loss, logits = model(**kw)
    assert logits.shape == (2, 25, model.decoder.config.vocab_size)
    err = (logits.float().cpu() - ref["logits"]).abs().max().item()
    assert err <= 1e-4, "fp32 logits differ from the oracle by %.3e (tolerance 1e-4)" % err
    assert abs(loss.item() - ref["loss"].item()) <= 1e-5 * max(1.0, abs(ref["loss"].item()))
    loss.backward()
    named = dict(model.named_parameters())
    worst = {}
    for k in keys:
        p = named.get(k)
        if p is None:                                   # aliased tensors are exposed under their first name only
            continue
        worst[k] = _rel(p.grad, ref["grads"][k])
    assert len(worst) >= 8
    assert max(worst.values()) <= 5e-4, worst
    assert _rel(feats.grad, ref["dfeats"]) <= 5e-4


def test_full_model_seq_len_256_train_mode_matches_oracle_under_engine_masks(full_fp32):
    """The shape AND mode the bench times (BASELINE configs[1]: seq_len 256 -> four 64-key chunks, positions 128-255, 293
    cross-attention keys; dropout on), on the 388 M-parameter model, 2 rows, fp32 parity mode: the engine's train-mode step
    against the oracle run with the masks the engine drew (155 sites; selfcheck.dropout_keep_masks).  Logits 1e-4, loss 1e-5,
    gradients across the depth of the network 5e-4 of each tensor's max."""
    import bench
    from gst_visdial_amd import selfcheck
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    model, _, sd, _, keys, _ = full_fp32
    O = _oracle()
    V = model.decoder.config.vocab_size
    batch = bench.synthetic_rows(2, 256, 37, 25, 2048, V, 8765, DEV)
    assert int(batch["enc_attention_mask"].sum(1).max()) > 200          # positions beyond 128 are really used
    model.train()
    try:
        model.zero_grad(set_to_none=True)
        feats = batch["enc_image_features"].clone().requires_grad_(True)
        loss, logits = model(**dict(batch, enc_image_features=feats))
        loss.backward()
        torch.cuda.synchronize()
        eng = model.engine
        table = selfcheck.dropout_keep_masks(eng)
        assert len(table) == 155 and len({v["site"] for v in eng.site_log.values()}) == 155
        masks = O.DropMasks(table)
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        cpu_batch = {k: v.cpu() for k, v in batch.items()}
        out, g, dfe = O.grads(sd, bert_base_enc_config(), bert_base_dec_config(), cpu_batch, keys, wrt_feats=True, train=masks)
        assert set(masks.used) == set(eng.site_log)
        for lab, v in eng.site_log.items():
            assert abs(v["p"] - masks.p_used[lab]) < 1e-12, lab
        err = (logits.float().cpu() - out["logits"]).abs().max().item()
        assert err <= 1e-4, "train-mode fp32 logits at seq_len 256 differ from the oracle by %.3e" % err
        assert abs(loss.item() - out["loss"].item()) <= 1e-5 * max(1.0, abs(out["loss"].item()))
        named = dict(model.named_parameters())
        worst = {k: _rel(named[k].grad, g[k]) for k in keys if k in named}
        assert len(worst) >= 8 and max(worst.values()) <= 5e-4, worst
        assert _rel(feats.grad, dfe) <= 5e-4
    finally:
        model.eval()
        model.zero_grad(set_to_none=True)


def test_full_size_dataparallel_wrapping_checkpoint_load_and_14_kwarg_call(full_fp32, tmp_path):
    """train_gen.py:293-295,118-135,324 at the reference's model size: nn.DataParallel(model, [0]), a checkpoint written and read
    the scripts' way (torch.save({'model_state_dict': ...}) / torch.load(map_location=device) / .module.load_state_dict), the 14
    keyword names incl. the dead inputs, lm_loss.mean().backward() -- logits / loss / gradients against the oracle."""
    import torch.nn as nn
    import bench
    model32, batch, sd, cpu_batch, keys, ref = full_fp32
    device = torch.device(DEV)
    model, params = bench.build_model(device, "fp32", seed=99)                    # different weights: the load must matter
    model = nn.DataParallel(model, params["gpu_ids"])
    ck = str(tmp_path / "full.ckpt")
    torch.save({"model_state_dict": sd, "iter_id": 0}, ck)
    state = torch.load(ck, map_location=device)
    res = model.module.load_state_dict(state["model_state_dict"])
    assert not res.missing_keys and not res.unexpected_keys
    del state
    model.eval()
    B, T = batch["enc_input_ids"].shape
    R = batch["enc_image_features"].shape[1]
    kw = dict(batch)
    kw.update(enc_image_target=torch.zeros(B, R, 1601, device=device), enc_image_label=torch.zeros(B, R, dtype=torch.long, device=device),
              enc_next_sentence_labels=torch.full((B,), -1, dtype=torch.long, device=device),
              enc_sep_indices=torch.zeros(B, 25, dtype=torch.long, device=device),
              enc_mlm_labels=torch.full((B, T), -1, dtype=torch.long, device=device))
    assert len(kw) == 14
    lm_loss, lm_scores = model(**kw)
    lm_loss = lm_loss.mean()
    assert (lm_scores.float().cpu() - ref["logits"]).abs().max().item() <= 1e-4
    assert abs(lm_loss.item() - ref["loss"].item()) <= 1e-5 * max(1.0, abs(ref["loss"].item()))
    lm_loss.backward()
    named = dict(model.module.named_parameters())
    worst = {k: _rel(named[k].grad, ref["grads"][k]) for k in keys if k in named}
    assert len(worst) >= 8 and max(worst.values()) <= 5e-4, worst


@pytest.mark.isolated
def test_full_size_questioner_decode_with_4gram_blocking_through_the_wrapper(full_fp32):
    """generate.py:124-142 at full size: q_model = nn.DataParallel(...); q_model(..., temperature=0.7, top_k, top_p=0.0,
    ngram_blocking_size=4).  Random-init contexts never repeat a 4-gram by themselves, so the ban is made to bite: the greedy
    sequence s of a first call (no ban) is planted, as question-segment tokens, into the PADDING of the context rows (history =
    enc_input_ids * (segments == 0), utils/decoding_utils.py:38-77 -- the attention mask plays no part in it, the encoder output
    of the real positions is unchanged).  With the ban on, the decode must follow s up to the first position whose token would
    complete a planted 4-gram and deviate exactly there; fp32 ids equal the oracle's decode on the same inputs (2 rows), and the
    bf16 engine at 16 rows shows the same property, eager and replayed."""
    import torch.nn as nn
    import bench
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    O = _oracle()
    model32, _, sd, _, _, _ = full_fp32
    device = torch.device(DEV)
    V = model32.decoder.config.vocab_size
    special = {0, 100, 101, 102, 103}

    def run(model, rows, ngram, Bn):
        kw = {k: rows[k] for k in ("enc_image_features", "enc_image_spatials", "enc_image_mask", "enc_input_ids", "enc_segments",
                                   "enc_attention_mask")}
        with torch.no_grad():
            return model(enc_image_target=None, enc_image_label=None, enc_next_sentence_labels=None, enc_sep_indices=None,
                         enc_mlm_labels=None, dec_input_ids=torch.full((Bn, 1), 101, dtype=torch.long, device=device),
                         dec_attention_mask=None, temperature=0.7, top_k=1, top_p=0.0, ngram_blocking_size=ngram,
                         uniforms=uni[Bn], **kw).clone()

    # the same uniforms for every call: bf16 logits TIE at the top now and then (8 mantissa bits over 30522 near-uniform values),
    # top_k = 1 keeps every tied value like the reference's `logits < kth` filter does, and the draw among them follows the uniform
    uni = {n: torch.rand(18, n, generator=torch.Generator().manual_seed(9)).clamp_min(1e-6).to(device) for n in (2, 16)}

    def plant(rows, s, k0=2):
        """tokens s[k0 .. k0+3] of every row -> the last four (padding) positions of its context, segment 0"""
        rows = {k: v.clone() for k, v in rows.items()}
        T = rows["enc_input_ids"].shape[1]
        assert float(rows["enc_attention_mask"][:, T - 4:].sum()) == 0          # really padding
        rows["enc_input_ids"][:, T - 4:] = s[:, k0:k0 + 4]
        rows["enc_segments"][:, T - 4:] = 0
        return rows

    for precision, Bn in (("fp32", 2), ("bf16", 16)):
        if precision == "fp32":
            model = model32
        else:
            model, _ = bench.build_model(device, "bf16", seed=7)
            model.load_state_dict({k: v.to(device) for k, v in sd.items()}, strict=True)
        model.eval()
        mode = model.params["mode"]
        model.params["mode"] = "vd_gen_val"
        wrapped = nn.DataParallel(model, [0])
        try:
            rows = bench.synthetic_rows(Bn, 256, 37, 25, 2048, V, 555, device)
            lens = rows["enc_attention_mask"].sum(1)
            keep = lens <= 250                                                   # rows with at least 6 padding positions
            if int(keep.sum()) < Bn:                                             # shorten the others: mask + ids of the tail go to 0
                rows["enc_input_ids"][:, 250:] = 0
                rows["enc_segments"][:, 250:] = 0
                rows["enc_attention_mask"][:, 250:] = 0
            s = run(wrapped, rows, 0, Bn)                                        # greedy, no ban
            ok = torch.tensor([not (special & set(s[b, 2:6].tolist())) for b in range(Bn)])
            assert int(ok.sum()) >= max(1, Bn // 2)                              # greedy rows that do not hit a special token early
            planted = plant(rows, s)
            same = run(wrapped, planted, 0, Bn)
            assert torch.equal(same, s)                                          # padding content does not change the model's outputs
            banned = run(wrapped, planted, 4, Bn)
            for b in range(Bn):
                if not ok[b]:
                    continue
                assert torch.equal(banned[b, :5], s[b, :5]), (precision, b)      # s[2..4] reproduced ...
                assert banned[b, 5].item() != s[b, 5].item(), (precision, b)     # ... and s[5] would complete the planted 4-gram
            if precision == "bf16":
                again = run(wrapped, planted, 4, Bn)                             # second call with these settings: hipGraph replay
                assert torch.equal(again, banned)
            else:
                cpu = {k: v.cpu() for k, v in planted.items()}
                cpu["dec_input_ids"] = torch.full((Bn, 1), 101, dtype=torch.long)
                torch.set_num_threads(min(os.cpu_count() or 1, 16))
                want, _ = O.sampling_decode(sd, bert_base_enc_config(), bert_base_dec_config(), cpu, 0.7, 1, 0.0, 4,
                                            draw=lambda p: p.argmax(-1, keepdim=True))
                assert torch.equal(banned.cpu(), want)
        finally:
            model.params["mode"] = mode


def _cc12m_loader_batch(n_dialogs, V, seed, zero_label_rate=0.4, T=256, R=37, U=25, F=2048):
    """A batch as the CC12M train loader emits it (dataloader/dataloader_cc12m_gen.py:193-200,238-266;
    utils/data_utils.py:89-101): per dialog [10, 1, L] text tensors, encoder inputs with 15 % of the tokens replaced by
    [MASK] = 103 (mask_prob 0.15; the MLM labels are dead on this path, the masked INPUTS are real), 15 % of the regions chosen
    and zeroed w.p. 0.9, and -- `-select_data` -- the label rows of the rounds whose answer perplexity is over the threshold zeroed."""
    import bench
    g = torch.Generator().manual_seed(seed)
    rows = bench.synthetic_rows(n_dialogs * 10, T, R, U, F, V, seed, "cpu")
    ids = rows["enc_input_ids"]
    special = (ids == 0) | (ids == 101) | (ids == 102)
    mask_tok = (torch.rand(ids.shape, generator=g) < 0.15) & ~special
    ids = torch.where(mask_tok, torch.full_like(ids, 103), ids)
    feats = rows["enc_image_features"][::10].clone()                        # one image per dialog
    chosen = torch.rand(n_dialogs, R, generator=g) < 0.15
    zero = chosen & (torch.rand(n_dialogs, R, generator=g) < 0.9)
    zero[:, 0] = False                                                      # (the global row is built from the boxes before masking)
    feats[zero] = 0
    labels = rows["dec_labels"].clone()
    dropped = torch.rand(n_dialogs * 10, generator=g) < zero_label_rate     # ppl >= threshold
    dropped[0] = True
    dropped[1] = False
    labels[dropped] = 0
    v = lambda t: t.view(n_dialogs, 10, 1, t.shape[-1])                     # noqa: E731
    batch = dict(enc_input_ids=v(ids), enc_segments=v(rows["enc_segments"]), enc_att_mask=v((ids != 0).float()),
                 dec_input_ids=v(rows["dec_input_ids"]), dec_att_mask=v(rows["dec_attention_mask"]), dec_labels=v(labels),
                 enc_image_feat=feats, enc_image_loc=rows["enc_image_spatials"][::10].clone(), enc_image_mask=torch.ones(n_dialogs, R))
    return batch, dropped, int(mask_tok.sum()), int(zero.sum())


def test_cc12m_self_training_step_full_size_select_data_masked_inputs(full_fp32):
    """BASELINE configs[4] on the 388 M-parameter model: a cc12m_train-shaped step through the step driver (train_gen.forward,
    train_gen.py:45-136) -- [MASK]-bearing encoder inputs, zeroed regions, `-select_data` zero-label rows filtered by the row
    sampler (train_gen.py:65-68).  fp32 parity mode, TRAIN mode (dropout on), 2 sampled rows: logits 1e-4 / loss 1e-5 /
    gradients 5e-4 against the oracle on exactly the sampled rows under the masks the engine drew; bf16 at the per-rank batch of
    10 rows: same rows, loss within the bf16 bar of the fp32 engine, every live gradient finite; a batch whose label rows are ALL
    zero fails loudly, as the reference's torch.multinomial does."""
    import bench
    from gst_visdial_amd import selfcheck, step
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    model, _, sd, _, keys, _ = full_fp32
    O = _oracle()
    V = model.decoder.config.vocab_size
    batch, dropped, n_mask, n_zero = _cc12m_loader_batch(10, V, 2468)
    assert n_mask > 1000 and n_zero > 20 and 20 < int(dropped.sum()) < 70
    dev = torch.device(DEV)
    params = model.params
    saved = (params["mode"], params["batch_size"])
    params["mode"], params["batch_size"], params["device"] = "cc12m_train", 2, dev
    model.train()
    try:
        # ---- the row sampler: only rows with a non-zero label row, same draw as the reference's multinomial on the host RNG
        gen = torch.Generator().manual_seed(77)
        rows, idx = step.select_rows(batch, params, None, gen)
        assert not dropped[idx].any() and (rows["dec_labels"].sum(-1) != 0).all()
        cand = O.candidate_rows(batch["dec_labels"].view(-1, 25))
        assert torch.equal(idx, torch.multinomial(cand, 2, replacement=True, generator=torch.Generator().manual_seed(77)))
        assert int((rows["enc_input_ids"] == 103).sum()) > 10               # the sampled rows really carry [MASK] tokens
        # ---- fp32, train mode, the two sampled rows vs the oracle under the engine's masks
        model.zero_grad(set_to_none=True)
        loss, logits = step.forward(model, batch, params, sample_indices=idx)
        loss.backward()
        torch.cuda.synchronize()
        masks = O.DropMasks(selfcheck.dropout_keep_masks(model.engine))
        cpu = dict(enc_image_features=rows["enc_image_feat"], enc_image_spatials=rows["enc_image_loc"], enc_image_mask=rows["enc_image_mask"],
                   enc_input_ids=rows["enc_input_ids"], enc_segments=rows["enc_segments"], enc_attention_mask=rows["enc_att_mask"],
                   dec_input_ids=rows["dec_input_ids"].clone(), dec_attention_mask=rows["dec_att_mask"], dec_labels=rows["dec_labels"])
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        out, g, _ = O.grads(sd, bert_base_enc_config(), bert_base_dec_config(), cpu, keys, wrt_feats=False, train=masks)
        assert set(masks.used) == set(model.engine.site_log)
        err = (logits.float().cpu() - out["logits"]).abs().max().item()
        assert err <= 1e-4, "cc12m step: fp32 train-mode logits differ from the oracle by %.3e" % err
        assert abs(loss.item() - out["loss"].item()) <= 1e-5 * max(1.0, abs(out["loss"].item()))
        named = dict(model.named_parameters())
        worst = {k: _rel(named[k].grad, g[k]) for k in keys if k in named}
        assert len(worst) >= 8 and max(worst.values()) <= 5e-4, worst
        # ---- an all-zero-label batch (every round over the perplexity threshold): loud failure, like torch.multinomial's
        empty = dict(batch, dec_labels=torch.zeros_like(batch["dec_labels"]))
        with pytest.raises(RuntimeError):
            step.forward(model, empty, params)
        # ---- bf16 at the per-rank batch of BASELINE configs[2] / [4]: 10 rows
        params["batch_size"] = 10
        gen = torch.Generator().manual_seed(78)
        rows10, idx10 = step.select_rows(batch, params, None, gen)
        assert not dropped[idx10].any()
        model.eval()                                                            # fp32 engine, dropout off: the bf16 run's yardstick
        with torch.no_grad():
            loss32, logits32 = step.forward(model, batch, params, sample_indices=idx10)
        m16, p16 = bench.build_model(dev, "bf16", seed=7)
        m16.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
        p16["mode"], p16["batch_size"], p16["device"] = "cc12m_train", 10, dev
        m16.eval()
        with torch.no_grad():
            loss16, logits16 = step.forward(m16, batch, p16, sample_indices=idx10)
        assert abs(loss16.item() - loss32.item()) <= 3e-2 * max(1.0, abs(loss32.item()))
        assert (logits16.float() - logits32.float()).abs().max().item() < 0.15
        m16.train()
        loss, _ = step.forward(m16, batch, p16, sample_indices=idx10)
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(loss).all()
        live = [(n, p) for n, p in m16.named_parameters() if p.grad is not None]
        assert len(live) > 200 and all(bool(torch.isfinite(p.grad).all()) for _, p in live)
        assert float(dict(live)["decoder.decoder.lm_head.decoder.weight"].grad.abs().max()) > 0
    finally:
        params["mode"], params["batch_size"] = saved
        model.eval()
        model.zero_grad(set_to_none=True)


def test_full_model_bf16_close_to_oracle(full_fp32):
    import bench
    model32, batch, sd, cpu_batch, keys, ref = full_fp32
    model, params = bench.build_model(torch.device(DEV), "bf16", seed=7)
    model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=True)
    model.eval()
    loss, logits = model(**batch)
    assert abs(loss.item() - ref["loss"].item()) <= 3e-2 * max(1.0, abs(ref["loss"].item()))
    err = (logits.float().cpu() - ref["logits"]).abs().max().item()
    assert err <= 0.15, err
    # ranking property: the argmax token of every supervised position agrees wherever the oracle's margin is clear
    lr, lg = ref["logits"], logits.float().cpu()
    top2 = lr.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.3
    assert (lg.argmax(-1)[clear] == lr.argmax(-1)[clear]).all()


@pytest.mark.parametrize("rows", [16, 10], ids=["configs1_16rows", "configs2_per_rank_10rows"])
def test_bench_shape_step_properties(rows):
    """configs[1] (16 rows, seq_len 256, bf16, dropout on) and configs[2]'s per-rank shape (global 80 on 8 GPUs = 10 rows per
    rank: M = 2560 / 370 / 250 instead of 4096 / 592 / 400 -- other tile grids, tails and split-K plans): finite loss near ln(V)
    for random weights, identical loss for identical (seed, offset) dropout state, loss changes when the dropout state
    advances, one AdamW step moves the loss."""
    import math
    import bench
    from gst_visdial_amd.optim import FusedAdamW
    model, params = bench.build_model(torch.device(DEV), "bf16", seed=3)
    model.train()
    V = model.decoder.config.vocab_size
    batch = bench.synthetic_rows(rows, 256, 37, 25, 2048, V, 99, DEV)
    model(**batch)                                        # builds the engine (flat buffers, rng state)
    st0 = model.engine.rng.state.clone()
    l0, logits = model(**batch)
    assert logits.shape == (rows, 25, V) and torch.isfinite(logits).all()
    assert abs(l0.item() - math.log(V)) < 1.5
    model.engine.rng.state.copy_(st0)
    l0b, _ = model(**batch)
    assert l0b.item() == l0.item()                       # same dropout masks -> bit-identical forward
    l1, _ = model(**batch)                                # offset advanced -> other masks
    assert l1.item() != l0.item()
    opt = FusedAdamW(model, lr=1e-3, warmup_steps=0, t_total=100)
    l1.backward()
    opt.step()
    opt.zero_grad()
    model.eval()
    l2, _ = model(**batch)
    assert torch.isfinite(l2) and l2.item() < l1.item() + 0.5


def test_full_size_candidate_chunk_properties(full_fp32):
    """BASELINE configs[3], scoring half at FULL size: one evaluate_gen chunk = 5 rounds x 100 candidates = 500 decoder rows
    (evaluate_gen.py:30,84) through score_candidates on the 388 M-parameter model, checked by size-independent properties:
      * encode-once scores == the expanded per-row path (every candidate row with its own encoder pass) on a slice of rows;
      * a sampled subset equals the CPU oracle's evaluate_gen arithmetic (expanded rows, fp32) within 2e-3;
      * permuting the candidates of a round permutes its scores and nothing else (rows are independent);
      * duplicated candidates score identically, and every score is a finite log-probability sum <= 0."""
    import bench
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    model, _, sd, _, _, _ = full_fp32
    O = _oracle()
    mode = model.params["mode"]
    model.params["mode"] = "vd_eval_val"
    try:
        E, G, T, U = 5, 100, 256, 25
        V = model.decoder.config.vocab_size
        rows = bench.synthetic_rows(E, T, 37, U, 2048, V, 77, DEV)
        cand = bench.synthetic_rows(E * G, 8, 37, U, 2048, V, 78, DEV)            # only its decoder side is used
        dec = cand["dec_input_ids"].clone()
        alen = (dec != 0).sum(-1)
        dec[torch.arange(E * G, device=DEV), alen.clamp(max=U - 1)] = 102         # eval rows end in [SEP] (dataloader_visdial_gen.py:230)
        dec[7] = dec[3]                                                            # a duplicated candidate inside round 0
        dmask = cand["dec_attention_mask"]
        enc = (rows["enc_image_features"], rows["enc_image_spatials"], rows["enc_image_mask"], rows["enc_input_ids"],
               rows["enc_segments"], rows["enc_attention_mask"])
        with torch.no_grad():
            fast = model.score_candidates(*enc, dec, dmask, G).clone()
            assert fast.shape == (E * G,) and torch.isfinite(fast).all() and (fast <= 0).all()
            assert fast[7].item() == fast[3].item()
            # per-row path on 12 rows spread over the rounds
            pick = torch.tensor([0, 3, 7, 99, 100, 157, 250, 299, 300, 420, 498, 499], device=DEV)
            rnd = pick // G
            slow = model.score_candidates(*(t[rnd] for t in enc), dec[pick], dmask[pick], 1)
            assert (fast[pick] - slow).abs().max().item() < 2e-4 * max(1.0, slow.abs().max().item())
            # candidate permutation inside every round
            perm = torch.stack([torch.randperm(G, generator=torch.Generator().manual_seed(r)) for r in range(E)]).to(DEV)
            idx = (torch.arange(E, device=DEV)[:, None] * G + perm).reshape(-1)
            shuffled = model.score_candidates(*enc, dec[idx], dmask[idx], G)
            assert (shuffled - fast[idx]).abs().max().item() < 1e-4
        # oracle on 4 of the rows (expanded the reference's way)
        sub = pick[[0, 4, 8, 11]]
        r4 = (sub // G).cpu()
        cpu = dict(enc_image_features=rows["enc_image_features"].cpu()[r4], enc_image_spatials=rows["enc_image_spatials"].cpu()[r4],
                   enc_image_mask=rows["enc_image_mask"].cpu()[r4], enc_input_ids=rows["enc_input_ids"].cpu()[r4],
                   enc_segments=rows["enc_segments"].cpu()[r4], enc_attention_mask=rows["enc_attention_mask"].cpu()[r4],
                   dec_input_ids=dec[sub].cpu().clone(), dec_attention_mask=dmask[sub].cpu(), dec_labels=None)
        unmutated = cpu["dec_input_ids"].clone()
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        out = O.model_forward(sd, bert_base_enc_config(), bert_base_dec_config(), cpu)
        ref = O.answer_scores(out["logits"], unmutated)
        assert (fast[sub].cpu() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    finally:
        model.params["mode"] = mode


@pytest.mark.isolated
def test_full_size_sampling_decode_replay_equals_eager_and_forced_prefix_matches_oracle():
    """BASELINE configs[3], decode half at FULL size (models/visual_dialog_model.py:74-120 on the 388 M-parameter model, bf16,
    16 rows x 18 sampled tokens, temperature 0.7 / top_k 7 as generate.py:138-141 sets them):
      * the hipGraph-replayed call (encoder graph + ONE token-loop graph) returns bit-for-bit the ids of the eagerly issued
        call under the same uniforms, also when the inputs are new (refreshed in place into the captured buffers);
      * KV-cached decode == teacher forcing: the call's last-position logits and the engine's teacher-forced pass over the
        sampled answer (rescore_sampled, which reuses the call's encoder states / cross-attention K/V) equal the CPU oracle's
        teacher-forced decoder on the same forced prefix, rows 0-1, at the bf16 closeness bar (0.15 on logits ~ +-10)."""
    import bench
    from gst_visdial_amd.config import bert_base_enc_config, bert_base_dec_config
    O = _oracle()
    dev = torch.device(DEV)
    model, params = bench.build_model(dev, "bf16", seed=5)
    model.eval()
    params["mode"] = "vd_gen_val"
    V = model.decoder.config.vocab_size
    Bn, steps = 16, 18

    def inputs(seed):
        d = bench.synthetic_rows(Bn, 256, 37, 25, 2048, V, seed, dev)
        return dict(enc_image_features=d["enc_image_features"], enc_image_spatials=d["enc_image_spatials"],
                    enc_image_mask=d["enc_image_mask"], enc_input_ids=d["enc_input_ids"], enc_segments=d["enc_segments"],
                    enc_attention_mask=d["enc_attention_mask"],
                    dec_input_ids=torch.full((Bn, 1), 101, dtype=torch.long, device=dev))
    args = dict(temperature=0.7, top_k=7, top_p=0.0, ngram_blocking_size=0)
    u = [torch.rand(steps, Bn, generator=torch.Generator().manual_seed(40 + i)).clamp_min(1e-6).to(dev) for i in range(2)]
    with torch.no_grad():
        params["amd_decode_graph"] = False
        eager = [model(uniforms=u[i], **args, **inputs(70 + i)).clone() for i in range(2)]
        params["amd_decode_graph"] = True
        first = model(uniforms=u[0], **args, **inputs(70)).clone()            # eager issue, then captures
        assert len(model.engine._decode_sessions) == 1
        replay0 = model(uniforms=u[0], **args, **inputs(70)).clone()          # replay, same inputs
        replay1 = model(uniforms=u[1], **args, **inputs(71)).clone()          # replay, new inputs + uniforms refreshed in place
        assert eager[0].shape == (Bn, steps)
        assert torch.equal(first, eager[0]) and torch.equal(replay0, eager[0]) and torch.equal(replay1, eager[1])
        assert not torch.equal(eager[0], eager[1])
        assert int((eager[0] >= V).sum()) == 0 and int((eager[0] < 0).sum()) == 0
        # ---- forced prefix vs the oracle (rows 0-1 of the second batch)
        ans = replay1.clone()
        # two rows that never sampled [SEP] (after it the returned ids are padded while the decode went on with raw tokens)
        sel = ((ans != 102) & (ans != 0)).all(1).nonzero().flatten()[:2]
        assert sel.numel() == 2
        last_logits = model.engine.last["decode_logits"][sel].float().cpu()   # raw logits of the last decode position
        full = torch.cat((torch.full((Bn, 1), 101, dtype=torch.long, device=dev), ans), dim=1)      # [CLS] + 18 sampled ids
        forced = full[:, :steps].clone()                                       # decoder input of the last position's pass
        _, tf_logits = model.engine.rescore_sampled(forced.clone(), None, loss_reduction=False)
        tf_logits = tf_logits[sel].float().cpu()
    kw = inputs(71)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    enc_cfg, dec_cfg = bert_base_enc_config(), bert_base_dec_config()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sel = sel.cpu()
    cpu = {k: v.cpu()[sel] for k, v in kw.items()}
    enc_t, enc_v = O.encoder_forward(sd, enc_cfg, cpu["enc_input_ids"], cpu["enc_segments"], cpu["enc_attention_mask"],
                                     cpu["enc_image_features"], cpu["enc_image_spatials"], cpu["enc_image_mask"], False)
    enc_h, enc_mask = O.vl_fusion(sd, enc_t, enc_v, cpu["enc_attention_mask"], cpu["enc_image_mask"], False)
    # the sampling branch feeds the raw prefix (no eos->pad mutation, no padding mask: visual_dialog_model.py:86-95)
    y = O.decoder_hidden(sd, dec_cfg, forced.cpu()[sel], None, enc_h, enc_mask, False)
    ref = O.lm_logits(sd, y)                                                   # [2, 18, V], position t from prefix[:t+1]
    assert (last_logits - ref[:, -1]).abs().max().item() < 0.15
    # rescore_sampled applies the labels=None conventions (eos -> pad in the decoder input) -- restate them for the oracle
    # (rescore_sampled applies the labels=None conventions -- eos -> pad in the decoder input --, a no-op on these rows)
    assert (tf_logits - ref).abs().max().item() < 0.15
