#!/bin/bash
# Round 5, GPU session 13: the LayerNorm-folded GEMMs (csrc/gemm_rows.hip) at the vision stream's width (H = 1024, NV = 4):
# op tests, the model-level parity tests, whole-step A/B of GSTVD_LN_FOLD_MAX_H = 768 (decoder sites only) / 1024, kernel stats.
export TMPDIR=/tmp; out=gpurun_out/r05_s13; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "layernorm_folded or layernorm_backward_folded" 2>&1 | tail -5 | tee $out/op_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q -k "two_stream or full_config or model or fused_update or round3" 2>&1 | tail -5 | tee $out/model_tests.log
bash tools/r04_step_ab.sh GSTVD_LN_FOLD_MAX_H 768 1024 2>&1 | tee $out/step_ab.txt
for v in 768 1024; do
  export GSTVD_LN_FOLD_MAX_H=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$v -- python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/prof$v.log 2>&1
  f=$(ls $out/prof$v/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_fold$v.csv; rm -rf $out/prof$v
done
python - <<'PY' | tee $out/kernel_delta.txt
import csv
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load('gpurun_out/r05_s13/kernel_stats_fold768.csv'), load('gpurun_out/r05_s13/kernel_stats_fold1024.csv')
steps = 18.0
for n in sorted(set(a) | set(b), key=lambda n: -abs(a.get(n, (0, 0))[1] - b.get(n, (0, 0))[1]))[:14]:
    ca, ta = a.get(n, (0, 0)); cb, tb = b.get(n, (0, 0))
    print('%-90s calls/step %6.1f -> %6.1f   ms/step %7.3f -> %7.3f' % (n[:90], ca / steps, cb / steps, ta / steps / 1e6, tb / steps / 1e6))
print('total kernel ms/step %.3f -> %.3f ; launches/step %.1f -> %.1f' % (sum(t for c, t in a.values()) / steps / 1e6, sum(t for c, t in b.values()) / steps / 1e6, sum(c for c, t in a.values()) / steps, sum(c for c, t in b.values()) / steps))
PY
