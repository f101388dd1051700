"""Round-4 GPU tests: the evaluate_gen gate (NDCG / MRR within 0.1 of the reference checkpoint) pinned on a real-shaped set --
a TRAINED tiny reference checkpoint, 8 dialogs x 10 rounds x 100 options with dense relevance (oracle/make_golden_r4.py ran the
reference itself) -- in the fp32 parity mode (ranks bit-exact) AND in the bf16 mode that ships (metrics within 0.1 point),
single process and sharded over two ranks; plus the boundary exactly as the reference's scripts use it (nn.DataParallel wrapping,
train_gen.py:293-295 / evaluate_gen.py:177-186 / generate.py:60-77)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("r@1", "r@5", "r@10", "mean", "mrr", "ndcg")


def sc():
    from gst_visdial_amd import selfcheck
    return selfcheck


def _reference_metrics(ev):
    m = {k: float(v) for k, v in zip(("r@1", "r@5", "r@10", "mean", "mrr"), ev["sparse"].tolist())}
    m["ndcg"] = float(ev["ndcg"].item())
    return m


def _points(m):
    """The scale the 0.1 gate is stated on: R@k, MRR, NDCG as percentages (x100), mean rank in ranks."""
    return {k: (m[k] if k == "mean" else 100.0 * m[k]) for k in KEYS}


def _evaluate(precision, batches=None, ev=None):
    from gst_visdial_amd import evaluate as EV
    s = sc()
    ev = ev if ev is not None else load_npz("tiny_evalset100.npz")
    model, params, _ = s.build_tiny_model(precision, DEV, mode="vd_eval_val", state_file="tiny_state_trained.npz")
    model.eval()
    batches = batches if batches is not None else s.evalset100_batches(ev)
    p = dict(params, device=torch.device(DEV), vd_version="1.0")
    scores = torch.cat([EV.score_batch(model, b, torch.device(DEV)).float().cpu() for b in batches])
    _, metrics = EV.evaluate(model, batches, p, mode="vd_eval_val")
    return scores, metrics


def test_eval_100_options_fp32_ranks_bit_exact_metrics_equal():
    """fp32 parity mode on the trained checkpoint: 8000 candidate scores within 1e-3 of the reference's, every rank equal, R@1/5/10,
    mean rank, MRR and NDCG equal to 1e-6 (evaluate_gen.py:94-118, utils/visdial_metrics.py:41-195)."""
    from gst_visdial_amd.metrics import scores_to_ranks
    ev = load_npz("tiny_evalset100.npz")
    scores, metrics = _evaluate("fp32", ev=ev)
    assert (scores - ev["scores"]).abs().max().item() < 1e-3
    assert torch.equal(scores_to_ranks(scores), ev["ranks"].long())
    ref = _reference_metrics(ev)
    for k in KEYS:
        assert abs(metrics[k] - ref[k]) < 1e-6, (k, metrics[k], ref[k])


def test_eval_100_options_bf16_metrics_within_a_tenth_of_a_point():
    """The mode that ships.  bf16 storage moves a candidate's score by ~1e-2 .. 1e-1 and flips near-tied neighbours; the gate is on
    the METRICS: each of R@1 / R@5 / R@10 / MRR / NDCG (x100) and the mean rank within 0.1 of the reference checkpoint's.
    The rank-flip rate is printed (DESIGN.md section 2 records it) and bounded."""
    from gst_visdial_amd.metrics import scores_to_ranks
    ev = load_npz("tiny_evalset100.npz")
    scores, metrics = _evaluate("bf16", ev=ev)
    ref = _reference_metrics(ev)
    err = (scores - ev["scores"]).abs()
    ranks, ref_ranks = scores_to_ranks(scores), ev["ranks"].long()
    moved = (ranks != ref_ranks).float().mean().item()
    gt = ev["in::gt_option_inds"].long()
    gt_rank = ranks.gather(-1, gt.unsqueeze(-1)).squeeze(-1)
    gt_rank_ref = ref_ranks.gather(-1, gt.unsqueeze(-1)).squeeze(-1)
    gt_moved = (gt_rank != gt_rank_ref).float().mean().item()
    big = (ranks - ref_ranks).abs().max().item()
    d = {k: _points(metrics)[k] - _points(ref)[k] for k in KEYS}
    print("\nbf16 evaluation vs the reference checkpoint (8 x 10 x 100): score err max %.3f mean %.4f | candidates whose rank moved "
          "%.4f (largest move %d places) | ground-truth ranks moved %.4f | metric deltas (points) %s"
          % (err.max().item(), err.mean().item(), moved, big, gt_moved, {k: round(v, 4) for k, v in d.items()}))
    for k in KEYS:
        assert abs(d[k]) <= 0.1 + 1e-9, (k, d[k], metrics[k], ref[k])
    assert err.max().item() < 0.5 and moved < 0.25 and big <= 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _eval_worker(rank, world, port, q, precision):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _, metrics = _evaluate(precision)
        q.put((rank, metrics))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as ex:          # noqa: BLE001
        import traceback
        q.put((rank, "ERROR: " + "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))[-1500:]))


@pytest.mark.isolated
def test_eval_100_options_bf16_sharded_over_two_ranks_equals_single_process():
    """evaluate() with a process group: the four 2-dialog batches go to ranks 0,1,0,1, no data-path exchange, metric state
    all-gathered once -- both ranks must report the single-process bf16 metrics exactly, i.e. within 0.1 point of the reference."""
    ev = load_npz("tiny_evalset100.npz")
    _, single = _evaluate("bf16", ev=ev)
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q, "bf16")) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
    ref = _reference_metrics(ev)
    for r in range(2):
        assert not isinstance(got[r], str), got[r]
        for k in KEYS:
            assert abs(got[r][k] - single[k]) < 1e-9, (r, k, got[r][k], single[k])
            assert abs(_points(got[r])[k] - _points(ref)[k]) <= 0.1 + 1e-9


# ------------------------------------------------------------------------------------------------ the boundary as the scripts use it
def _script_kwargs(g, dev, B, T):
    """The 14 keyword arguments of train_gen.py:118-133, incl. the ones the enc_dec path ignores (dead inputs the reference's loop
    still passes: image target / label, next-sentence labels, sep indices, MLM labels)."""
    return dict(enc_image_features=g["in::enc_image_features"].to(dev), enc_image_spatials=g["in::enc_image_spatials"].to(dev),
                enc_image_mask=g["in::enc_image_mask"].to(dev),
                enc_image_target=torch.zeros(B, g["in::enc_image_features"].shape[1], 1601, device=dev),
                enc_image_label=torch.zeros(B, g["in::enc_image_features"].shape[1], dtype=torch.long, device=dev),
                enc_next_sentence_labels=torch.full((B,), -1, dtype=torch.long, device=dev),
                enc_input_ids=g["in::enc_input_ids"].to(dev), enc_segments=g["in::enc_segments"].to(dev),
                enc_sep_indices=torch.zeros(B, 25, dtype=torch.long, device=dev),
                enc_mlm_labels=torch.full((B, T), -1, dtype=torch.long, device=dev),
                enc_attention_mask=g["in::enc_attention_mask"].to(dev), dec_input_ids=g["in::dec_input_ids"].clone().to(dev),
                dec_attention_mask=g["in::dec_attention_mask"].to(dev), dec_labels=g["in::dec_labels"].to(dev))


def test_dataparallel_wrapping_as_train_gen_evaluate_gen_and_generate_use_it(tmp_path):
    """Literally what the reference's scripts do with the model (train_gen.py:293-295,118-135,324; evaluate_gen.py:177-186;
    generate.py:60-77,124-142,185-211): build, alias the embeddings, .to(device), nn.DataParallel(model, [0]), load a checkpoint
    through `.module.load_state_dict(ckpt['model_state_dict'])` (strict), call with the 14 keyword names, `lm_loss.mean()
    .backward()`, flip `.module.params['mode']` between decode / 'train' (the perplexity trick) -- against the golden numbers the
    reference itself produced for the tiny config."""
    import torch.nn as nn
    from gst_visdial_amd.modules import VisualDialogEncoder, VisualDialogDecoder, EncoderDecoderModel
    s = sc()
    base, params, cfg = s.build_tiny_model("fp32", DEV, mode="vd_train")         # only for its config files / params dict
    device = torch.device(DEV)
    params = dict(params, device=device, gpu_ids=[0], mode="vd_train")
    enc, dec = VisualDialogEncoder(params), VisualDialogDecoder(params)
    dec.decoder.bert.embeddings = enc.bert_pretrained.bert.embeddings          # train_gen.py:293 / generate.py:62
    model = EncoderDecoderModel(params, enc, dec).to(device)
    model = nn.DataParallel(model, params["gpu_ids"])                           # train_gen.py:295
    ck = str(tmp_path / "visdial_dialog_encoder.ckpt")
    torch.save({"model_state_dict": load_npz("tiny_state.npz")}, ck)            # the reference's checkpoint dict (train_gen.py:341-349)
    state = torch.load(ck, map_location=device)                                  # generate.py:66
    missing = model.module.load_state_dict(state["model_state_dict"])            # strict by default, as in the scripts
    assert not missing.missing_keys and not missing.unexpected_keys
    g = load_npz("tiny_train.npz")
    B, T = g["in::enc_input_ids"].shape
    model.eval()                                                                # the golden vectors are dropout-free
    lm_loss, lm_scores = model(**_script_kwargs(g, device, B, T))
    lm_loss = lm_loss.mean()                                                    # train_gen.py:134-135
    assert (lm_scores.float().cpu() - g["logits"]).abs().max().item() < 1e-4
    assert abs(lm_loss.item() - g["loss"].item()) < 1e-5
    lm_loss.backward()                                                          # train_gen.py:324
    named = dict(model.module.named_parameters())
    checked = 0
    for k in g:
        if k.startswith("grad::") and k[6:] in named:
            ref = g[k]
            got = named[k[6:]].grad
            assert got is not None, k
            assert (got.float().cpu() - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-6), k
            checked += 1
    assert checked >= 25
    assert all(k.startswith("module.") for k in model.state_dict())              # what train_gen.py:343 strips via .module
    # ---- generate.py:124-142: the questioner's decode call (ngram_blocking_size=4) through the wrapper
    dc = load_npz("tiny_decode.npz")
    model.module.params["mode"] = "vd_gen_val"
    kw = _script_kwargs(g, device, B, T)
    for k in ("dec_labels",):
        kw.pop(k)
    kw.update(enc_image_target=None, enc_image_label=None, enc_next_sentence_labels=None, enc_sep_indices=None, enc_mlm_labels=None,
              dec_input_ids=torch.full((B, 1), 101, dtype=torch.long, device=device))
    with torch.no_grad():
        ids4 = model(temperature=0.7, top_k=1, top_p=0.0, ngram_blocking_size=4, **kw)
        ids0 = model(temperature=0.7, top_k=1, top_p=0.0, ngram_blocking_size=0, **kw)
    assert ids4.shape == (B, 18) and ids0.shape == (B, 18)
    from oracle import vd_oracle as O                                            # the checker
    cpu = {k[4:]: v.clone() for k, v in g.items() if k.startswith("in::")}
    cpu["dec_input_ids"] = torch.full((B, 1), 101, dtype=torch.long)
    sd = load_npz("tiny_state.npz")
    for ng, got in ((4, ids4), (0, ids0)):
        want, _ = O.sampling_decode(sd, cfg["enc"], cfg["dec"], cpu, 0.7, 1, 0.0, ng, draw=lambda p: p.argmax(-1, keepdim=True))
        assert torch.equal(got.cpu(), want), (ng, got.cpu(), want)
    # ---- generate.py:183-211: the perplexity trick -- mode 'train', no labels, loss_reduction=False, [B*18] token losses
    ans_ids = ids0.clone()
    model.module.params["mode"] = "train"
    kw["dec_input_ids"] = ans_ids
    kw["dec_attention_mask"] = (ans_ids != 0).float()
    with torch.no_grad():
        loss, _ = model(loss_reduction=False, **kw)
    assert loss.shape == (B * 18,)
    cpu["dec_input_ids"] = ids0.cpu().clone()
    cpu["dec_attention_mask"] = (ids0.cpu() != 0).float()
    cpu["dec_labels"] = None
    ref = O.model_forward(sd, cfg["enc"], cfg["dec"], cpu, loss_reduction=False)
    assert (loss.float().cpu() - ref["loss"].reshape(-1)).abs().max().item() < 1e-4
    assert torch.equal(kw["dec_input_ids"].cpu(), cpu["dec_input_ids"])          # the in-place eos -> pad mutation happened on both
    model.module.params["mode"] = "vd_gen_val"                                   # generate.py:211 restores the mode
