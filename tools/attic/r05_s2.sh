#!/bin/bash
# Round 5, GPU session 2: one-pass attention backward (tests + step A/B + kernel averages), XCD placement probe
export TMPDIR=/tmp; out=gpurun_out/r05_s2; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
build/xcd_probe 5352 > $out/xcd_probe.txt 2>&1; build/xcd_probe 1024 >> $out/xcd_probe.txt 2>&1; cat $out/xcd_probe.txt
python3 -X faulthandler -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py -x -q -m gpu -p no:cacheprovider -k "attention or adamw" > $out/attn_tests.log 2>&1; echo "rc=$?" >> $out/attn_tests.log
tail -12 $out/attn_tests.log
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -4 $out/gpu_tests_full.log
for rnd in 1 2; do for v in 0 1; do
  r=$(GSTVD_ATTN_ONEPASS=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_ATTN_ONEPASS=$v round $rnd: ms_per_step value = $r" | tee -a $out/onepass_ab.txt
done; done
for v in 0 1; do
  GSTVD_ATTN_ONEPASS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$v -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof$v.log 2>&1
  f=$(ls $out/stats$v/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_onepass$v.csv; grep -i "attn" $out/kernel_stats_onepass$v.csv | cut -c1-200
  rm -rf $out/stats$v
done
