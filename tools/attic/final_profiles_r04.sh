#!/bin/bash
# Round 4: every measurement committed under profiles/r04_* from ONE script (run on the GPU box; outputs under gpurun_out/final,
# copied to profiles/ by tools/collect_profiles.py r04).   bash tools/final_profiles_r04.sh
export TMPDIR=/tmp; out=gpurun_out/final; rm -rf $out; mkdir -p $out
# 1. the full GPU test-suite, whole log kept (the driver's command)
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -3 $out/gpu_tests_full.log
# 2. the contract line with the driver's flags + per-kernel event breakdown (default side measurements: configs[3] decode / score,
#    CPU baseline by BASELINE.md's procedure, fp32 parity-mode timing, PCIe-inclusive rate); wall time of the whole command
t0=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 --breakdown-json $out/breakdown_events.json > $out/bench_stdout.log 2> $out/bench_stderr.log
echo "python3 bench.py --steps 20 --warmup 5 (the driver's flags, all side measurements on): $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" > $out/bench_wall.txt
tail -1 $out/bench_stdout.log > $out/bench_n1.json; cat $out/bench_wall.txt; cut -c1-260 $out/bench_n1.json
# 3. rocprofv3 --stats of the same command (kernel averages must agree with roofline.avg_launch_us)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/prof_bench.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats_bench.csv; rm -rf $out/stats
head -8 $out/kernel_stats_bench.csv | cut -c1-150
# 4. HBM traffic per kernel family inside the step (separate --pmc passes) and MFMA pipe utilisation of the step's GEMM shapes
bash tools/pmc_bench.sh > $out/pmc.log 2>&1; cp gpurun_out/pmc3/summary.json $out/pmc_traffic.json; rm -rf gpurun_out/pmc3
bash tools/pmc_mfma.sh > $out/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.json $out/pmc_mfma_util.json; rm -rf gpurun_out/pmc_mfma
# 5. timeline of a steady-state replayed step + the compressed kernel trace
bash tools/trace_step.sh --no-eval-decode > $out/timeline.txt 2>&1; cp gpurun_out/trace/kernel_trace.csv.gz $out/kernel_trace_steps.csv.gz
# 6. the same GEMMs against the vendor BLAS (speed-of-light reference only) + K slopes, NT64 off / on
( for v in 0 1; do echo "== GSTVD_GEMM_NT64=$v"; GSTVD_GEMM_NT64=$v python3 tools/gemm_bench.py all lib; GSTVD_GEMM_NT64=$v python3 tools/nt_study.py; done ) 2>/dev/null | grep -v amdgpu > $out/gemm_vs_vendor_blas.txt
python3 tools/cold_probe.py 2>/dev/null | grep -v amdgpu > $out/cold_operands.txt
# 7. launch floor: the same graph at 2 rows x 32 tokens, kernels per step
( echo "== bench.py at 2 rows x 32 tokens (same number of graph nodes)"; python3 bench.py --rows-per-gpu 2 --seq-len 32 --steps 30 --warmup 3 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown 2>/dev/null | tail -1 | cut -c1-200
  bash tools/trace_step.sh --rows-per-gpu 2 --seq-len 32 --no-eval-decode > /dev/null 2>&1; python3 tools/by_kernel.py ) > $out/launch_floor.txt 2>&1
# 8. the bench's launch paths: plain --gpus 2 on this 1-GPU box must refuse; self-launched 2-rank control flow (all ranks on cuda:0,
#    gloo, eager) incl. the child-process legs; the 1-rank RCCL path at both row counts + legs in fresh children
( echo "== python3 bench.py --gpus 2 on a 1-GPU box"; python3 bench.py --gpus 2 --steps 3 --warmup 1; echo "exit code $? (must be non-zero)"
  echo "== GSTVD_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 (self-launched; legs as fresh child processes)"
  GSTVD_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown 2>/dev/null | tail -1; echo "exit code $?" ) > $out/bench_launch_paths.txt 2>&1
for rows in 16 10; do GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 2 --rows-per-gpu $rows --grad-compress bf16 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>/dev/null | tail -1 > $out/bench_force_dist_rows$rows.json; done
GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --legs on --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown 2>/dev/null | tail -1 > $out/bench_force_dist_legs.json
python3 bench.py --steps 10 --warmup 2 --rows-per-gpu 10 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d 2>/dev/null | tail -1 > $out/bench_n1_rows10.json
tail -4 $out/bench_launch_paths.txt | cut -c1-300; cut -c1-200 $out/bench_force_dist_legs.json
# 9. whole-step A/B records of this round's switches (two interleaved rounds each)
( bash tools/r04_step_ab.sh GSTVD_GEMM_NT64 0 1; bash tools/r04_step_ab.sh GSTVD_FUSE_UPDATE 0 1 ) > $out/step_ab.txt 2>&1; cat $out/step_ab.txt
# 10. weight gradients + AdamW: two launches against the one launch with the update in its epilogue, stand-alone (C++ over the C ABI)
( build/fused_update_bench 24 12; build/fused_update_bench 48 24 ) > $out/fused_update_bench.txt 2>&1; tail -4 $out/fused_update_bench.txt
