#!/bin/bash
# Kernel trace of a few replayed train steps (rocprofv3 --kernel-trace), kept as gpurun_out/trace/kernel_trace.csv.gz so that
# tools/trace_report.py can be run on it again and again without a GPU.   usage: tools/trace_step.sh [extra bench.py args]
export TMPDIR=/tmp; out=gpurun_out/trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d "$@" > $out/bench.log 2>&1
f=$(ls $out/*/*kernel_trace.csv | head -1)
gzip -c "$f" > $out/kernel_trace.csv.gz
rm -rf $out/*/
tail -1 $out/bench.log | cut -c1-300
python3 tools/trace_report.py $out/kernel_trace.csv.gz
