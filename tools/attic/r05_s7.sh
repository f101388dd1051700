#!/bin/bash
# Round 5, GPU session 7: keep bits A/B in one session; sharded-update tests after the packing change
export TMPDIR=/tmp; out=gpurun_out/r05_s7; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
python3 -X faulthandler -m pytest tests/test_round5_gpu.py tests/test_dp_gpu.py -x -q -m gpu -p no:cacheprovider > $out/sel_tests.log 2>&1; echo "rc=$?" >> $out/sel_tests.log
tail -5 $out/sel_tests.log
for rnd in 1 2 3; do for v in 0 1; do
  r=$(GSTVD_ATTN_KEEP_BITS=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_ATTN_KEEP_BITS=$v round $rnd: ms_per_step value = $r" | tee -a $out/keep_bits_ab.txt
done; done
for sh in off on; do
  GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 2 --grad-compress bf16 --shard-update $sh --legs off $LEAN 2>$out/dist_$sh.err | tail -1 > $out/bench_force_dist_shard_$sh.json
  python3 -c "import json; d=json.load(open('$out/bench_force_dist_shard_$sh.json')); print('shard-update $sh:', d['ms_per_step'], d['config'].get('optimizer_update'), d['config'].get('hip_graph'))" | tee -a $out/shard_ab.txt
done
