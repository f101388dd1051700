#!/bin/bash
# Round 5, GPU session 3: one-pass attention backward v2, native n-gram ban, 64-tile ring depth A/B
export TMPDIR=/tmp; out=gpurun_out/r05_s3; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
python3 -X faulthandler -m pytest tests/test_ops_gpu.py tests/test_round5_gpu.py tests/test_round2_gpu.py -x -q -m gpu -p no:cacheprovider -k "attention or adamw or ngram or sampl or decode" > $out/sel_tests.log 2>&1; echo "rc=$?" >> $out/sel_tests.log
tail -12 $out/sel_tests.log
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -4 $out/gpu_tests_full.log
for rnd in 1 2; do for v in 0 1; do
  r=$(GSTVD_ATTN_ONEPASS=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_ATTN_ONEPASS=$v round $rnd: ms_per_step value = $r" | tee -a $out/onepass_ab.txt
done; done
for rnd in 1 2; do for v in 8 4 3; do
  r=$(GSTVD_GEMM64_NS=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_GEMM64_NS=$v round $rnd: ms_per_step value = $r" | tee -a $out/gemm64_ns_ab.txt
done; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof.log 2>&1
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -30 $out/kernel_stats.csv | cut -c1-180; rm -rf $out/stats
python3 tools/eval_decode_bench.py > $out/eval_decode.json 2>$out/eval_decode.err; cat $out/eval_decode.json
