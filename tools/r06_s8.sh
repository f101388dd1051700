#!/bin/bash
# Round 6, GPU session 4 (re-run as session 8 with the XCD-local hand-off): the Linear with the LayerNorm in its epilogue (gstvd_gemm_ln_epi): op tests, engine A/B, whole-step A/B
# (GSTVD_FUSE_LN_EPI 0 / 1, two interleaved rounds), stand-alone kernel times, eager issue after the raw-stream accessor.
export TMPDIR=/tmp; out=gpurun_out/r06_s8; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity"
timeout 900 python3 -X faulthandler -m pytest tests/test_round6_gpu.py -x -q -m gpu -p no:cacheprovider > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
grep -n "Error\|error\|passed\|failed\|rc=\|assert" $out/new_tests.log | head -30
timeout 400 bash tools/step_ab.sh GSTVD_FUSE_LN_EPI 0 1 2>&1 | tee $out/ln_epi_step_ab.txt
for v in 0 1; do
  GSTVD_FUSE_LN_EPI=$v python3 bench.py --steps 10 --warmup 3 $LEAN 2>/dev/null | tail -1 > $out/bench_lnepi$v.json
  python3 -c "
import json
d=json.load(open('$out/bench_lnepi$v.json')); r=d['roofline']
print('GSTVD_FUSE_LN_EPI=$v: %.3f ms/step; launches %s; coattn_frac %s; hbm:' % (d['ms_per_step'], r['launching_calls_per_step'], r['coattn_frac']), [(h['kernel'], h['launches'], h['ms_per_step'], h['frac']) for h in r['hbm']])
print('   breakdown:', json.dumps(d.get('kernel_breakdown_ms')))
" | tee -a $out/ln_epi_bench.txt
done
python3 bench.py --steps 10 --warmup 3 --graph off $LEAN --no-breakdown 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('--graph off: %.3f ms/step, eager_host_issue_ms_per_step %s' % (d['ms_per_step'], d['config']['eager_host_issue_ms_per_step']))" | tee $out/graph_off.txt
python3 - <<'PY' | tee gpurun_out/r06_s8/ln_epi_kernel_times.txt
import torch, sys
sys.path.insert(0, ".")
from gst_visdial_amd import ops as o
DEV = "cuda:0"; bf = torch.bfloat16
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for (M, N, K) in ((4096, 768, 768), (4096, 768, 1024), (4096, 768, 3072), (2560, 768, 768), (2560, 768, 3072)):
    x = torch.randn(M, K, device=DEV).to(bf); w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(bf); b = torch.randn(N, device=DEV)
    res = torch.randn(M, N, device=DEV).to(bf); gamma = torch.ones(N, device=DEV); beta = torch.zeros(N, device=DEV)
    g, y = torch.empty(M, N, device=DEV, dtype=bf), torch.empty(M, N, device=DEV, dtype=bf)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    rng = o.Rng(DEV, seed=1)
    kw = dict(mode=o.LN_RESID, dtype=o.BF16, M=M, H=N, gamma=gamma, beta=beta, mean=mean, rstd=rstd, eps=1e-12, x=g, res=res, y=y, p_pre=0.1, site_pre=2, rng=rng)
    ws = o.gemm_ln_epi_ws(M, N, torch.device(DEV))
    tg = t(lambda: o.gemm(x, w, g, M, N, K, bias=b)); tl = t(lambda: o.ln_fwd(**kw))
    tb = t(lambda: (o.gemm(x, w, g, M, N, K, bias=b), o.ln_fwd(**kw)))
    tf = t(lambda: o.gemm_ln_epi(x, w, g, kw, N, K, ws, bias=b))
    ta = t(lambda: o.gemm_ln_epi(x, w, g, kw, N, K, ws, bias=b, xcd_local=False))
    print("%5d x %4d x %4d: gemm %.1f us, ln_fwd %.1f us, back to back %.1f us | one launch: XCD-local %.1f us, agent scope %.1f us (error flag %s, mapping ok %s)" % (M, N, K, tg, tl, tb, tf, ta, o.gemm_ln_epi_error(ws), o.xcd_mapping_ok(torch.device(DEV))))
PY
