#!/bin/bash
# Round 6, GPU session 5: N > 1 tail -- direct bf16 payload out of the weight-gradient launch: tests, forced-dist
# step with and without it (all-reduce and sharded update), timeline of the forced-dist step, RCCL soak.
export TMPDIR=/tmp; out=gpurun_out/r06_s5; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity --no-breakdown --legs off"
timeout 900 python3 -X faulthandler -m pytest tests/test_round6_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
grep -n "Error\|error\|passed\|failed\|rc=\|assert" $out/new_tests.log | head -30
for rnd in 1 2; do for d in 0 1; do for sh in off on; do
  GSTVD_BENCH_DIRECT_BF16=$d GSTVD_FORCE_DIST=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --grad-compress bf16 --shard-update $sh $LEAN 2> $out/err.txt | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('round $rnd direct_bf16=$d shard-update $sh: %.3f ms/step hip_graph %s' % (d['ms_per_step'], d['config']['hip_graph']))
except Exception as e:
    print('round $rnd direct_bf16=$d shard-update $sh: FAILED', repr(e))
" | tee -a $out/direct_bf16_ab.txt
done; done; done
grep -v "^\[W\|amdgpu.ids" $out/err.txt | tail -3
