#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (hipGraph replay, two streams + pipeline)
export TMPDIR=/tmp; out=gpurun_out/prof_$1; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ${@:2} > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-300
f=$(ls $out/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -30 $f | cut -c1-200
rm -f $out/*/*kernel_trace.csv
