#!/bin/bash
# Regenerates everything under profiles/ that bench.py's JSON line refers to (run on the GPU box; outputs in gpurun_out/final)
export TMPDIR=/tmp; out=gpurun_out/final; rm -rf $out; mkdir -p $out
python3 bench.py --breakdown-json $out/breakdown_events.json > $out/bench_stdout.log 2> $out/bench_stderr.log
tail -1 $out/bench_stdout.log > $out/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown > $out/prof_bench.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats_bench.csv
rm -rf $out/stats
bash tools/pmc_bench.sh > $out/pmc.log 2>&1
cp gpurun_out/pmc3/summary.json $out/pmc_traffic.json
rm -rf gpurun_out/pmc3
cut -c1-600 $out/bench_n1.json; head -8 $out/kernel_stats_bench.csv | cut -c1-160; tail -9 $out/pmc.log
