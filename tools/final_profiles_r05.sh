#!/bin/bash
# Round 5: every measurement committed under profiles/r05_* that describes the FINAL build, from ONE script (run on the GPU box; outputs
# under gpurun_out/final, copied to profiles/ by tools/collect_profiles.py r05).   bash tools/final_profiles_r05.sh
# (The A/B records behind the round's decisions -- group order, attention, ring depths, tile widths, sc1 stores, keep bits -- were
# written by tools/r05_s1.sh .. r05_s8.sh while the decisions were made; section 9 below repeats the switches that stayed.)
export TMPDIR=/tmp; out=gpurun_out/final; rm -rf $out; mkdir -p $out
LEAN="--no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d"
# 1. the full GPU test-suite, whole log kept (the driver's command)
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
tail -3 $out/gpu_tests_full.log
# 2. the contract line with the driver's flags + per-kernel event breakdown (all side measurements on); wall time of the whole command
t0=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 --breakdown-json $out/breakdown_events.json > $out/bench_stdout.log 2> $out/bench_stderr.log
echo "python3 bench.py --steps 20 --warmup 5 (the driver's flags, all side measurements on): $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s wall" > $out/bench_wall.txt
tail -1 $out/bench_stdout.log > $out/bench_n1.json; cat $out/bench_wall.txt; cut -c1-260 $out/bench_n1.json
# 3. rocprofv3 --stats of the same command (kernel averages must agree with roofline.avg_launch_us)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 $LEAN > $out/prof_bench.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats_bench.csv; rm -rf $out/stats
head -8 $out/kernel_stats_bench.csv | cut -c1-150
# 4. HBM traffic per kernel family inside the step (separate --pmc passes) and MFMA pipe utilisation of the step's GEMM shapes
bash tools/pmc_bench.sh > $out/pmc.log 2>&1; cp gpurun_out/pmc3/summary.json $out/pmc_traffic.json; rm -rf gpurun_out/pmc3
bash tools/pmc_mfma.sh > $out/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.json $out/pmc_mfma_util.json; rm -rf gpurun_out/pmc_mfma
# 5. timeline of a steady-state replayed step + the compressed kernel trace + alone / overlapped durations per kernel family
bash tools/trace_step.sh --no-eval-decode > $out/timeline.txt 2>&1; cp gpurun_out/trace/kernel_trace.csv.gz $out/kernel_trace_steps.csv.gz
python3 tools/overlap_stats.py $out/kernel_trace_steps.csv.gz > $out/overlap_stats.txt 2>&1
# 6. the same GEMMs against the vendor BLAS (speed-of-light reference only) + K slopes
( python3 tools/gemm_bench.py all lib; python3 tools/nt_study.py ) 2>/dev/null | grep -v amdgpu > $out/gemm_vs_vendor_blas.txt
# 7. launch floor: the same graph at 2 rows x 32 tokens, kernels per step
( echo "== bench.py at 2 rows x 32 tokens (same number of graph nodes)"; python3 bench.py --rows-per-gpu 2 --seq-len 32 --steps 30 --warmup 3 $LEAN 2>/dev/null | tail -1 | cut -c1-200
  bash tools/trace_step.sh --rows-per-gpu 2 --seq-len 32 --no-eval-decode > /dev/null 2>&1; python3 tools/by_kernel.py ) > $out/launch_floor.txt 2>&1
# 8. the bench's launch paths on a 1-GPU box + the N > 1 code path on a 1-rank RCCL group (all-reduce and sharded update, legs in fresh children)
( echo "== python3 bench.py --gpus 2 on a 1-GPU box"; python3 bench.py --gpus 2 --steps 3 --warmup 1; echo "exit code $? (must be non-zero)"
  echo "== GSTVD_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 (self-launched; legs as fresh child processes)"
  GSTVD_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 $LEAN 2>/dev/null | tail -1; echo "exit code $?" ) > $out/bench_launch_paths.txt 2>&1
for rows in 16 10; do GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 2 --rows-per-gpu $rows --grad-compress bf16 --legs off $LEAN 2>/dev/null | tail -1 > $out/bench_force_dist_rows$rows.json; done
GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --legs on $LEAN 2>/dev/null | tail -1 > $out/bench_force_dist_legs.json
python3 bench.py --steps 10 --warmup 2 --rows-per-gpu 10 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d 2>/dev/null | tail -1 > $out/bench_n1_rows10.json
tail -4 $out/bench_launch_paths.txt | cut -c1-300; cut -c1-200 $out/bench_force_dist_legs.json
# 9. whole-step A/B records of the switches that stayed (two interleaved rounds each)
( bash tools/r04_step_ab.sh GSTVD_GROUP_ORDER 0 1 3; bash tools/r04_step_ab.sh GSTVD_ATTN_ONEPASS 0 1; bash tools/r04_step_ab.sh GSTVD_GEMM64_NS 8 3
  bash tools/r04_step_ab.sh GSTVD_ATTN_KEEP_BITS 0 1; bash tools/r04_step_ab.sh GSTVD_ATTN_BWD_ORDER 1 0; bash tools/r04_step_ab.sh GSTVD_FUSE_UPDATE 0 1 ) > $out/step_ab.txt 2>&1; cat $out/step_ab.txt
# 10. eval / decode side measurements (decode with and without the questioner's 4-gram ban)
python3 tools/eval_decode_bench.py > $out/eval_decode.json 2> /dev/null; cat $out/eval_decode.json
# 11. 10 clean fresh-process runs of the 1-rank RCCL captured step, all-reduce and sharded update
( for i in 1 2 3 4 5; do for sh in off on; do
    GSTVD_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --grad-compress bf16 --shard-update $sh --legs off $LEAN 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i shard-update $sh: rc ok', d['ms_per_step'], 'ms, hip_graph', d['config']['hip_graph'], d['config']['capture_quiesce'])" || echo "run $i shard-update $sh: FAILED"
  done; done ) > $out/rccl_soak.txt 2>&1; tail -4 $out/rccl_soak.txt
