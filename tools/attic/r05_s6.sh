#!/bin/bash
# Round 5, GPU session 6: dropout keep bits forward -> one-pass backward
export TMPDIR=/tmp; out=gpurun_out/r05_s6; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
python3 -X faulthandler -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py -x -q -m gpu -p no:cacheprovider -k "attention or keep_bits" > $out/attn_tests.log 2>&1; echo "rc=$?" >> $out/attn_tests.log
tail -8 $out/attn_tests.log
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -4 $out/gpu_tests_full.log
for rnd in 1 2 3; do
  r=$(python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "keep bits round $rnd: ms_per_step value = $r" | tee -a $out/bits_ab.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof.log 2>&1
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; grep -i "attn" $out/kernel_stats.csv | cut -c1-200; rm -rf $out/stats
