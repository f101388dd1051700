#!/bin/bash
# N>1 code path on ONE GPU, repeatedly: (a) the 1-rank RCCL process-group test (eager + hipGraph-captured pipeline step) in
# fresh processes, (b) bench.py with GSTVD_FORCE_DIST=1 (RCCL collectives, graded slices, graph capture) at 16 and 10 rows/GPU.
# Prints one line per run; any non-zero exit code is a failure.  usage: tools/rccl_soak.sh [runs=10]
n=${1:-10}; out=gpurun_out/r2/rccl_soak; mkdir -p $out; fail=0
for i in $(seq 1 $n); do
  GSTVD_TEST_CHILD=1 timeout 300 python3 -X faulthandler -m pytest "tests/test_model_gpu.py::test_pipeline_with_rccl_collectives_single_rank_group" -x -q -p no:cacheprovider > $out/test_$i.log 2>&1
  rc=$?; echo "rccl 1-rank captured step, run $i: rc=$rc"; [ $rc -ne 0 ] && fail=1
done
for rows in 16 10 16 10; do
  GSTVD_FORCE_DIST=1 timeout 600 python3 bench.py --steps 10 --warmup 2 --rows-per-gpu $rows --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/bench_$rows.json 2> $out/bench_$rows.err
  rc=$?; echo "bench FORCE_DIST rows/gpu=$rows: rc=$rc $(tail -1 $out/bench_$rows.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], "hip_graph", d["config"]["hip_graph"], d["config"]["grad_allreduce_dtype"])' 2>/dev/null)"; [ $rc -ne 0 ] && fail=1
done
python3 bench.py --steps 10 --warmup 2 --rows-per-gpu 10 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bench N=1 rows/gpu=10:", d["value"], d["ms_per_step"])'
grep -l "drained\|timer" $out/*.log 2>/dev/null | head -1
echo "soak failures: $fail"
exit $fail
