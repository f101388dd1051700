#!/bin/bash
# round 4: the bench's launch paths on a 1-GPU box.  (1) default N=1 line; (2) plain `--gpus 2` must refuse; (3) self-launched
# 2-rank control flow on one GPU (gloo validation mode) incl. the child legs; (4) 1-rank RCCL path + legs in fresh children
out=gpurun_out/r04_bench_check; mkdir -p $out
echo "== (1) N=1 default" ; timeout 900 python3 bench.py --steps 20 --warmup 5 > $out/n1.json 2> $out/n1.err; echo "rc=$?"; tail -c 3000 $out/n1.json
echo "== (2) --gpus 2 on a 1-GPU box"; timeout 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $out/gpus2.out 2> $out/gpus2.err; echo "rc=$? (must be non-zero)"; tail -3 $out/gpus2.err
echo "== (3) self-launched 2 ranks, one-GPU validation mode (gloo, eager), legs on"
GSTVD_BENCH_ONE_GPU=1 timeout 1500 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d > $out/onegpu2.json 2> $out/onegpu2.err; echo "rc=$?"; tail -c 2500 $out/onegpu2.json; tail -5 $out/onegpu2.err
echo "== (4) 1-rank RCCL path, legs on"
GSTVD_FORCE_DIST=1 timeout 1500 python3 bench.py --steps 10 --warmup 3 --legs on --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown > $out/force_dist_legs.json 2> $out/force_dist_legs.err; echo "rc=$?"; tail -c 2500 $out/force_dist_legs.json; tail -5 $out/force_dist_legs.err
