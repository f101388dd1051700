export TMPDIR=/tmp; out=gpurun_out/final; mkdir -p $out
t0=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 --breakdown-json $out/breakdown_events.json > $out/bench_stdout.log 2> $out/bench_stderr.log
echo "python3 bench.py --steps 20 --warmup 5 (the driver's flags, all side measurements on): $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" > $out/bench_wall.txt
tail -1 $out/bench_stdout.log > $out/bench_n1.json; cat $out/bench_wall.txt; cut -c1-200 $out/bench_n1.json
