#!/bin/bash
# Round 6, GPU session 9: the decoder's cross-K/V projection in two launches (engine.CKV_SPLIT): test, full-size parity tests, step A/B, timeline
export TMPDIR=/tmp; out=gpurun_out/r06_s9; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity --no-breakdown"
timeout 900 python3 -X faulthandler -m pytest tests/test_round6_gpu.py tests/test_full_config_gpu.py tests/test_round3_gpu.py -x -q -m gpu -p no:cacheprovider -k "cross_kv or full_size or bench_path or oracle" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -n "Error\|error\|passed\|failed\|rc=\|assert" $out/tests.log | head -20
for rnd in 1 2 3; do for v in 0 2; do
  r=$(GSTVD_BENCH_CKV_SPLIT=$v timeout 300 python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "CKV_SPLIT=$v round $rnd: ms_per_step value = $r" | tee -a $out/ckv_split_ab.txt
done; done
for v in 0 2; do GSTVD_BENCH_CKV_SPLIT=$v GSTVD_FORCE_DIST=1 timeout 300 python3 bench.py --steps 20 --warmup 5 --grad-compress bf16 --legs off $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced-dist CKV_SPLIT=$v:', d['ms_per_step'], 'ms')" | tee -a $out/ckv_split_ab.txt; done
timeout 300 bash tools/trace_step.sh --no-eval-decode --no-rows-sensitivity > $out/timeline.txt 2>&1; head -12 $out/timeline.txt | cut -c1-250
