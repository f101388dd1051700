#!/bin/bash
# Round 5, GPU session 18b: step A/B and kernel time of GSTVD_GROUP_ORDER 1 / 3 / 4 (the first run of r05_s18.sh lost its step numbers to
# a bug in the one-liner that printed them; its FETCH / WRITE figures stand)
export TMPDIR=/tmp; out=gpurun_out/r05_s18; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
for rnd in 1 2 3; do for v in 1 3 4; do
  r=$(GSTVD_GROUP_ORDER=$v python3 bench.py --steps 30 --warmup 5 $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "GSTVD_GROUP_ORDER=$v round $rnd: ms_per_step value = $r" | tee -a $out/order_ab2.txt
done; done
for v in 1 3 4; do
  GSTVD_GROUP_ORDER=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$v -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof$v.log 2>&1
  f=$(ls $out/stats$v/*/*kernel_stats.csv | head -1); echo "GSTVD_GROUP_ORDER=$v: $(grep grouped_adamw $f | cut -d, -f8-12)" | tee -a $out/order_ab2.txt
  rm -rf $out/stats$v
done
