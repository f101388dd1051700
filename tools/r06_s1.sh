#!/bin/bash
# Round 6, GPU session 1 (evidence before kernel work):
#  1. what the vendor BLAS kernel IS on the three large-N NT shapes where it is ahead: kernel name (the Tensile name encodes the
#     macro tile, depthU, wave-group shape, LDS-direct / prefetch flags), grid, workgroup, LDS, VGPR / AGPR, scratch -- from a
#     rocprofv3 kernel trace of tools/gemm_bench.py (ours and the vendor's, same operands, same process);
#  2. rows sensitivity: the same bench at 32 and 64 rows per GPU (never `value`): does coattn_frac move with the row count?
#  3. the box's own 16-row reference line for the round's A/Bs.
export TMPDIR=/tmp; out=gpurun_out/r06_s1; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d"
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/gemm_bench.py main lib > $out/gemm_bench.txt 2> $out/gemm_bench.err
python3 tools/vendor_kernel_config.py $(ls $out/trace/*/*kernel_trace.csv | head -1) > $out/vendor_kernel_config.txt 2>&1
rm -rf $out/trace
cat $out/vendor_kernel_config.txt | cut -c1-400
for rows in 16 32 64; do
  python3 bench.py --steps 10 --warmup 3 --rows-per-gpu $rows $LEAN > $out/bench_rows$rows.log 2> $out/bench_rows$rows.err
  tail -1 $out/bench_rows$rows.log > $out/bench_rows$rows.json
  python3 - $out/bench_rows$rows.json $rows <<'PY' | tee -a $out/rows_sensitivity.txt
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("rows %s: ms_per_step %.3f value %.1f step_frac %s coattn_frac %s all_gemm_tflops %s launches %s" % (
        sys.argv[2], d["ms_per_step"], d["value"], r.get("step_frac"), r.get("coattn_frac"), r.get("all_gemm_tflops"), r.get("launching_calls_per_step")))
    print("   coattn:", json.dumps(r.get("coattn")))
except Exception as e:
    print("rows %s: FAILED %r" % (sys.argv[2], e))
PY
done
tail -5 $out/bench_rows64.err
