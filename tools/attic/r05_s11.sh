#!/bin/bash
# Round 5, GPU session 11: L2 prefetch of the optimizer state in the fused weight-gradient + AdamW epilogue (ADAMW_L2_PREFETCH), A/B by
# swapping two builds of the library (lib/libgstvd_hip_pf{0,1}.so, made by `make EXTRA=-DADAMW_L2_PREFETCH=0|1`).
export TMPDIR=/tmp; out=gpurun_out/r05_s11; rm -rf $out; mkdir -p $out
L=gst_visdial_amd/lib
cp $L/libgstvd_hip_pf1.so $L/libgstvd_hip.so
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or adamw or update or group" 2>&1 | tail -5 | tee $out/fused_tests.log
for r in 1 2; do for v in 0 1; do
  cp $L/libgstvd_hip_pf$v.so $L/libgstvd_hip.so
  echo "prefetch=$v round $r stand-alone:" | tee -a $out/ab.txt
  build/fused_update_bench 24 12 1 2>&1 | tail -3 | tee -a $out/ab.txt
  python bench.py --steps 60 --warmup 10 > $out/bench_pf${v}_$r.json 2>$out/bench.err
  python - $out/bench_pf${v}_$r.json <<'PY' | tee -a $out/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('  step: ms_per_step %.3f value %.1f roofline.achieved %s' % (d['ms_per_step'], d['value'], d['roofline'].get('achieved')))
PY
done; done
for v in 0 1; do
  cp $L/libgstvd_hip_pf$v.so $L/libgstvd_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$v -- python3 bench.py --steps 15 --warmup 3 > $out/prof$v.log 2>&1
  f=$(ls $out/prof$v/*/*kernel_stats.csv | head -1); head -4 $f | tee -a $out/ab.txt; cp $f $out/kernel_stats_pf$v.csv; rm -rf $out/prof$v
done
cp $L/libgstvd_hip_pf1.so $L/libgstvd_hip.so
#  the line above left the L2-PREFETCH build -- measured slower, not adopted -- in place of the default library on the box it ran on;
# harmless there (a gpurun box is thrown away), but do not run this record against a checkout you keep: rebuild with `make -C gst_visdial_amd/csrc` afterwards.
