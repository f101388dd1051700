#!/bin/bash
# Round 5, GPU session 17: ABI 6 -- top-p and unbounded top-k inside gstvd_sample_topk: op / decode tests, sampling kernel time by setting.
export TMPDIR=/tmp; out=gpurun_out/r05_s17; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "sampl or top_p or decode or ngram or cabi" 2>&1 | tail -6 | tee $out/sample_tests.log
python - <<'PY' 2>&1 | tee $out/sample_kernel_us.txt
import torch, time
from gst_visdial_amd import ops
dev = 'cuda:0'
B, V = 16, 30522
g = torch.Generator().manual_seed(0)
for dtype in (torch.bfloat16,):
    logits = (torch.randn(B, 30528, generator=g) * 2).to(dev).to(dtype)[:, :V]
    u = torch.rand(B, generator=g).clamp_min(1e-6).to(dev)
    out = torch.zeros(B, dtype=torch.long, device=dev)
    for top_k, top_p in [(7, 0.0), (64, 0.0), (65, 0.0), (1000, 0.0), (0, 0.9), (7, 0.9), (0, 0.5), (1000, 0.95)]:
        for _ in range(5): ops.sample_topk(logits, 0.7, top_k, u, out, None, top_p=top_p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.sample_topk(logits, 0.7, top_k, u, out, None, top_p=top_p)
        e1.record(); torch.cuda.synchronize()
        print('sample_topk bf16 16 x 30522  top_k %5d top_p %.2f: %.1f us per launch' % (top_k, top_p, e0.elapsed_time(e1) * 1000 / 50))
PY
