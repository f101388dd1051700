#!/bin/bash
# Round 6, GPU session 2: the hygiene build (zero-scratch epilogues, rejected switches gone) -- full GPU suite, the driver's bench
# command with the new side records (rows sensitivity, rocprof figure beside the live one, 5-step CPU baseline), scratch sizes of
# the step's kernels as the hardware reports them, and the vendor kernel of the cross-K/V shape (session 1's filter missed it).
export TMPDIR=/tmp; out=gpurun_out/r06_s2; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-rows-sensitivity"
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -12 $out/gpu_tests_full.log
t0=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 > $out/bench_stdout.log 2> $out/bench_stderr.log
echo "python3 bench.py --steps 20 --warmup 5: $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" | tee $out/bench_wall.txt
tail -1 $out/bench_stdout.log > $out/bench_n1.json; cut -c1-300 $out/bench_n1.json; tail -3 $out/bench_stderr.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s2/bench_n1.json"))
print("rows_sensitivity:", json.dumps(d["config"].get("rows_sensitivity")))
r = d["roofline"]; print("roofline:", {k: r.get(k) for k in ("frac", "avg_launch_us", "avg_launch_us_rocprof", "frac_at_rocprof_duration", "coattn_frac", "step_frac", "launching_calls_per_step")})
print("cpu_baseline:", json.dumps(d["cpu_baseline"]))
PY
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 3 --warmup 2 $LEAN --no-breakdown > $out/prof_bench.log 2>&1
python3 - <<'PY' | tee gpurun_out/r06_s2/scratch_by_kernel.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r06_s2/trace/*/*kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:90]
    a = agg.setdefault(k, [0, 0, r["VGPR_Count"], r["LDS_Block_Size"]])
    a[0] += 1; a[1] = max(a[1], int(r["Scratch_Size"]))
print("kernels with scratch > 0 (Scratch_Size column of the kernel trace, bytes per lane):")
n = 0
for k, (c, s, v, l) in agg.items():
    if s > 0: print("  %5d launches  scratch %4d  VGPR %s  %s" % (c, s, v, k)); n += 1
print("  none" if n == 0 else "  (%d kernels)" % n)
PY
rm -rf $out/trace
rocprofv3 --kernel-trace --output-format csv -d $out/trace2 -- python3 tools/gemm_bench.py main lib > $out/gemm_vs_vendor_blas.txt 2> /dev/null
python3 - <<'PY' | tee gpurun_out/r06_s2/big_kernels.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r06_s2/trace2/*/*kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < 100: continue
    k = (r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])
    a = agg.setdefault(k, [0, 0.0, r]); a[0] += 1; a[1] += d
for (n, g, w), (c, t, r) in agg.items():
    print("%4d x %7.1f us  grid %s wg %s LDS %s VGPR %s AGPR %s scratch %s  %s" % (c, t / c, g, w, r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], n[:420]))
PY
rm -rf $out/trace2
cut -c1-200 $out/gemm_vs_vendor_blas.txt
