#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r06_s3; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity --no-breakdown --legs off"
python3 -X faulthandler -m pytest tests/test_round6_gpu.py -x -q -m gpu -p no:cacheprovider > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
grep -n "Error\|passed\|failed\|rc=" $out/new_tests.log | head -20
for g in segmented; do for sh in off on; do
  GSTVD_FORCE_DIST=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --grad-compress bf16 --shard-update $sh --graph $g $LEAN 2> $out/dist_${g}_$sh.err | tail -1 > $out/dist_${g}_$sh.json
  python3 -c "
import json
try:
    d=json.load(open('$out/dist_${g}_$sh.json')); c=d['config']
    print('--graph $g shard-update $sh: %.3f ms/step, hip_graph %s, graphs/step %s, slices %s, update: %s' % (d['ms_per_step'], c.get('hip_graph'), c.get('graphs_per_step'), c.get('gradient_slices_per_step'), c.get('optimizer_update')))
except Exception as e:
    print('--graph $g shard-update $sh: FAILED', repr(e))
" | tee -a $out/segmented_ab.txt
  grep -v "^\[W\|amdgpu.ids" $out/dist_${g}_$sh.err | tail -4 | cut -c1-300
done; done
