#!/bin/bash
# Round 6, GPU session 3: segmented replay (tests + forced-dist bench: whole-step graph vs segmented vs eager, all-reduce and sharded
# update), host-time profile of the eager step (what a C-side submit would have to remove).
export TMPDIR=/tmp; out=gpurun_out/r06_s3; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity --no-breakdown --legs off"
python3 -X faulthandler -m pytest tests/test_round6_gpu.py tests/test_dp_gpu.py -x -q -m gpu -p no:cacheprovider > $out/new_tests.log 2>&1; echo "rc=$?" >> $out/new_tests.log
tail -25 $out/new_tests.log
for g in auto segmented off; do for sh in off on; do
  GSTVD_FORCE_DIST=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --grad-compress bf16 --shard-update $sh --graph $g $LEAN 2> $out/dist_${g}_$sh.err | tail -1 > $out/dist_${g}_$sh.json
  python3 -c "
import json
try:
    d=json.load(open('$out/dist_${g}_$sh.json')); c=d['config']
    print('--graph $g shard-update $sh: %.3f ms/step, hip_graph %s, graphs/step %s, slices %s, update: %s' % (d['ms_per_step'], c.get('hip_graph'), c.get('graphs_per_step'), c.get('gradient_slices_per_step'), c.get('optimizer_update')))
except Exception as e:
    print('--graph $g shard-update $sh: FAILED', repr(e))
" | tee -a $out/segmented_ab.txt
  tail -2 $out/dist_${g}_$sh.err | cut -c1-300
done; done
python3 tools/host_profile.py > $out/host_profile.txt 2>&1; cat $out/host_profile.txt | cut -c1-170
