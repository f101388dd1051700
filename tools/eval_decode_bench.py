#!/usr/bin/env python3
"""Side measurements for SURVEY 8(f-1)/(f-2) at the full model size (bf16, 1 GPU), synthetic inputs:
  * evaluate_gen scoring: 5 rounds x 100 candidates per chunk -- encode-once (score_candidates) vs the reference's
    expanded 500-row batch through model(...) + answer-score gather
  * sampling decode: 16 rows, 18 steps, KV cache -- eager issue vs hipGraph replay."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

dev = torch.device("cuda", 0)
model, params = bench.build_model(dev, "bf16", seed=1)
model.eval()
V = model.decoder.config.vocab_size
out = {}

def note(msg):
    print("[eval_decode_bench] " + msg, file=sys.stderr, flush=True)

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# ---- evaluate_gen chunk: 5 rounds x 100 options
R_, O_ = 5, 100
b = bench.synthetic_rows(R_, 256, 37, 25, 2048, V, 5, dev)
cand = bench.synthetic_rows(R_ * O_, 256, 37, 25, 2048, V, 6, dev)
with torch.no_grad():
    params["mode"] = "vd_eval_val"
    def once():
        return model.score_candidates(b["enc_image_features"], b["enc_image_spatials"], b["enc_image_mask"], b["enc_input_ids"],
                                      b["enc_segments"], b["enc_attention_mask"], cand["dec_input_ids"].clone(),
                                      cand["dec_attention_mask"], O_)
    def expanded():
        rep = lambda x: x.repeat_interleave(O_, 0)
        return model(enc_image_features=rep(b["enc_image_features"]), enc_image_spatials=rep(b["enc_image_spatials"]),
                     enc_image_mask=rep(b["enc_image_mask"]), enc_input_ids=rep(b["enc_input_ids"]), enc_segments=rep(b["enc_segments"]),
                     enc_attention_mask=rep(b["enc_attention_mask"]), dec_input_ids=cand["dec_input_ids"].clone(),
                     dec_attention_mask=cand["dec_attention_mask"], dec_labels=None, loss_reduction=False)
    note("scoring: encode-once")
    t1 = timeit(once)
    note("scoring: expanded batch")
    t2 = timeit(expanded, n=2, warm=1)
    out["eval_scoring_500_candidates"] = {"encode_once_ms": round(t1 * 1e3, 2), "expanded_batch_ms": round(t2 * 1e3, 2),
                                          "candidates_per_s_encode_once": round(500 / t1, 1), "speedup": round(t2 / t1, 1)}
    # ---- decode
    params["mode"] = "vd_gen_val"
    d = bench.synthetic_rows(16, 256, 37, 25, 2048, V, 7, dev)
    kw = dict(enc_image_features=d["enc_image_features"], enc_image_spatials=d["enc_image_spatials"], enc_image_mask=d["enc_image_mask"],
              enc_input_ids=d["enc_input_ids"], enc_segments=d["enc_segments"], enc_attention_mask=d["enc_attention_mask"],
              dec_input_ids=torch.full((16, 1), 101, dtype=torch.long, device=dev), temperature=0.7, top_k=7, top_p=0.0, ngram_blocking_size=0)
    params["amd_decode_graph"] = False
    note("decode: eager")
    te = timeit(lambda: model(**kw), n=3, warm=1)
    params["amd_decode_graph"] = True
    note("decode: hipGraph")
    tg = timeit(lambda: model(**kw), n=5, warm=2)
    out["sampling_decode_16rows_18steps"] = {"eager_ms": round(te * 1e3, 2), "hipgraph_ms": round(tg * 1e3, 2),
                                             "rows_per_s_hipgraph": round(16 / tg, 1), "speedup": round(te / tg, 2)}
    # ---- perplexity re-score of the sampled answer (generate.py:183-211): decode state reused vs encoder + decoder again
    from gst_visdial_amd.generate import answer_perplexity
    enc_kw = {k: v for k, v in kw.items() if k.startswith("enc_")}
    def sample_then_ppl(reuse):
        a = model(**kw)
        return answer_perplexity(model, enc_kw, a, reuse_decode_state=reuse)
    note("perplexity: decode state reused")
    tr = timeit(lambda: sample_then_ppl(True), n=5, warm=2)
    note("perplexity: full re-run")
    tf = timeit(lambda: sample_then_ppl(False), n=5, warm=2)
    out["answer_perplexity_16rows"] = {"reusing_decode_state_ms": round((tr - tg) * 1e3, 2), "full_rerun_ms": round((tf - tg) * 1e3, 2),
                                       "sample_plus_ppl_ms": round(tr * 1e3, 2)}
    kw["ngram_blocking_size"] = 4                             # question generation (generate.py:141): n-gram ban on the device
    note("decode: hipGraph, 4-gram ban")
    tn = timeit(lambda: model(**kw), n=5, warm=2)
    out["sampling_decode_16rows_18steps_ngram4"] = {"hipgraph_ms": round(tn * 1e3, 2), "rows_per_s_hipgraph": round(16 / tn, 1)}
print(json.dumps(out))
