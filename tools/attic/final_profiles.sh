#!/bin/bash
# Regenerates every measurement committed under profiles/ for a round from ONE script (run on the GPU box; outputs go to
# gpurun_out/final, copy them to profiles/<round>_* afterwards with tools/collect_profiles.py).
#   bash tools/final_profiles.sh
export TMPDIR=/tmp; out=gpurun_out/final; rm -rf $out; mkdir -p $out
# 1. the full GPU test-suite, whole log kept (the driver's command)
python3 -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > $out/gpu_tests_full.log 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests_full.log
# 2. the contract line + per-kernel event breakdown (default flags: CPU baseline, fp32 parity-mode timing, PCIe-inclusive rate)
python3 bench.py --breakdown-json $out/breakdown_events.json > $out/bench_stdout.log 2> $out/bench_stderr.log
tail -1 $out/bench_stdout.log > $out/bench_n1.json
# 3. rocprofv3 --stats of the same command (kernel averages must agree with roofline.avg_launch_us)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d > $out/prof_bench.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats_bench.csv; rm -rf $out/stats
# 4. serialized per-(kernel, grid) durations
bash tools/prof_serial.sh > $out/prof_serial.log 2>&1; cp gpurun_out/prof_serial/by_kernel_grid.json $out/serialized_by_kernel_grid.json
# 5. HBM traffic per kernel family inside the step (separate --pmc passes)
bash tools/pmc_bench.sh > $out/pmc.log 2>&1; cp gpurun_out/pmc3/summary.json $out/pmc_traffic.json; rm -rf gpurun_out/pmc3
# 6. MFMA pipe utilisation of the step's GEMM shapes incl. the connection-layer ones
bash tools/pmc_mfma.sh > $out/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.json $out/pmc_mfma_util.json
# 7. timeline of a STEADY-STATE replayed step (round 3: the median of the replays that are followed by another replay; round 2's
#    tools/timeline.sh reported the last step of the burst, which has 0.6-0.9 ms of idle time the others do not have)
bash tools/trace_step.sh > $out/timeline.txt 2>&1; cp gpurun_out/trace/kernel_trace.csv.gz $out/kernel_trace_steps.csv.gz
# 8. GEMM K-loop study: K slope + ablations + in-kernel clock (diagnostic builds, env-selected)
( for ab in 0 1 7 2; do GSTVD_DIAG_ABLATE=$ab python3 tools/clock_probe.py 3072; done
  for ab in 0 1 2 6; do GSTVD_GEMM256_NIU=4 GSTVD_DIAG_ABLATE=$ab python3 tools/kslope.py nt 4096 4096 | sed "s/^/ABLATE=$ab /"; done
  for st in 0 4; do GSTVD_GEMM256_NIU=4 GSTVD_GEMM_ST=$st python3 tools/kslope.py nt 4096 4096 | sed "s/^/ST=$st /"; done
  python3 tools/kslope.py nt 4096 768 64; python3 tools/kslope.py nn 4096 768 64
  python3 tools/gemm_bench.py all; python3 tools/write_floor.py ) > $out/gemm_study.txt 2>/dev/null
rm -f gst_visdial_amd/lib/libgstvd_hip_diag.so      # (built by the probes above through tools/diag_lib.py; never shipped)
# 9. host input path next to the replayed step
( for m in pinned_async pinned pageable; do python3 tools/h2d_probe.py --mode=$m | tail -5; done ) > $out/h2d_probe.txt 2>/dev/null
# 10. eval / decode side measurements
python3 tools/eval_decode_bench.py > $out/eval_decode.json 2> /dev/null
# 10b. the decode token step kernel by kernel (rocprofv3 kernel trace of a replayed call), the sampling kernel by top_k, and the
#      attention kernels at the step's six shapes
rocprofv3 --kernel-trace --output-format csv -d $out/dec_trace -- python3 tools/decode_debug.py > /dev/null 2>&1
( python3 tools/decode_timeline.py $out/dec_trace; python3 tools/sample_probe.py 2>/dev/null | grep top_k; bash tools/attn_shapes.sh 2>/dev/null | grep "^attn" ) > $out/decode_attention_kernels.txt 2>&1; rm -rf $out/dec_trace
# 10c. (round 3) sweeps that chose the backward pipeline's slice lists are kept as profiles/r03_chunk_sweep.txt (tools/chunk_sweep.sh);
#      here: the defaults against round 2's lists, N = 1 and the 1-rank RCCL path
rm -f gpurun_out/chunk_sweep.txt
tools/chunk_sweep.sh "22,27,27,27,27,192" "192" > /dev/null 2>&1
GSTVD_PIPE_UPDATE_STREAM=0 tools/chunk_sweep.sh "192" > /dev/null 2>&1
GSTVD_FORCE_DIST=1 tools/chunk_sweep.sh "22,27,27,27,27,96,96,32,16" "128,96,96,32,16" > /dev/null 2>&1
cp gpurun_out/chunk_sweep.txt $out/slice_defaults_check.txt
# 10d. evidence behind the round-3 decisions: vendor-BLAS reference timings + K slopes, row scaling of the decoder's kernels,
#      cost of an in-kernel grid barrier (tools/attic/gridbar.hip.txt, built to build/gridbar by the caller)
( python3 tools/gemm_bench.py all lib; python3 tools/nt_study.py ) 2>/dev/null | grep -v amdgpu > $out/gemm_vs_vendor_blas.txt
bash tools/row_split_probe.sh > $out/row_split_probe.txt 2>&1
[ -x build/gridbar ] && timeout 60 build/gridbar > $out/grid_barrier.txt 2>&1
# 10e. (round 3) what bounds the step besides its FLOPs: the same graph at 2 rows x 32 tokens (launch-count floor) with its
#      per-kernel durations and the host time of hipGraphLaunch; two concurrent chains of small GEMMs on real streams (no graph);
#      attention kernels by batch rows / with dropout off + their SQ counters; ring depth A/B of the 128-tile GEMMs
( echo "== bench.py at 2 rows x 32 tokens (same number of graph nodes)"; python3 bench.py --rows-per-gpu 2 --seq-len 32 --steps 30 --warmup 3 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown 2>/dev/null | tail -1 | cut -c1-200
  bash tools/trace_step.sh --rows-per-gpu 2 --seq-len 32 > /dev/null 2>&1; python3 tools/by_kernel.py
  echo "== hipGraphLaunch host time vs GPU span"; bash tools/launch_probe.sh 2 32 2>&1 | tail -5; bash tools/launch_probe.sh 16 256 2>&1 | tail -5
  echo "== two chains of dependent 400x768x768 GEMMs from a C++ loop (mode 0: one full-size chain; 1: two half-size chains on two streams; 2: the same on one stream)"
  [ -x build/chain_bench ] && build/chain_bench 400 96 && build/chain_bench 400 400 ) > $out/launch_floor.txt 2>&1
( echo "== text self-attention 12 heads x 256 x 256 x 64 by batch rows"; for b in 1 4 8 16 32; do python3 tools/attn_probe.py $b 12 256 256 64 0 0.1 50 2>/dev/null | tail -1; done
  echo "== dropout off"; python3 tools/attn_probe.py 16 12 256 256 64 0 0.0 50 2>/dev/null | tail -1
  bash tools/pmc_attn.sh 2>&1 | grep -v amdgpu | tail -16 ) > $out/attention_study.txt 2>&1
( for i in 1 2 3; do for ns in 5 3; do echo -n "GSTVD_GEMM128_NS=$ns: "; GSTVD_GEMM128_NS=$ns python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d --no-breakdown 2>/dev/null | tail -1 | cut -c60-100; done; done ) > $out/ring_depth_ab.txt 2>&1
cut -c1-400 $out/bench_n1.json; tail -3 $out/gpu_tests_full.log; head -6 $out/kernel_stats_bench.csv | cut -c1-150
# 11. the N>1 code path on one GPU (1-rank RCCL group: collectives, graded slices, bf16 payload, graph capture), both row counts
for rows in 16 10; do GSTVD_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 2 --rows-per-gpu $rows --grad-compress bf16 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>/dev/null | tail -1 > $out/bench_force_dist_rows$rows.json; done
python3 bench.py --steps 10 --warmup 2 --rows-per-gpu 10 --no-cpu-baseline --no-eval-decode --no-fp32 --no-h2d 2>/dev/null | tail -1 > $out/bench_n1_rows10.json
cut -c1-300 $out/bench_force_dist_rows10.json
