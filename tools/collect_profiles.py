#!/usr/bin/env python3
"""Copy what tools/final_profiles.sh (+ the soak / repro runs) left under gpurun_out/ into profiles/<round>_* (tracked).
usage: collect_profiles.py r03"""
import os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "final"), os.path.join(root, "profiles")
names = {"bench_n1.json": "bench_n1.json", "breakdown_events.json": "breakdown_events.json", "kernel_stats_bench.csv": "kernel_stats_bench.csv",
         "serialized_by_kernel_grid.json": "serialized_by_kernel_grid.json", "pmc_traffic.json": "pmc_traffic.json",
         "pmc_mfma_util.json": "pmc_mfma_util.json", "timeline.txt": "timeline.txt", "gemm_study.txt": "gemm_kloop_study.txt",
         "h2d_probe.txt": "h2d_probe.txt", "eval_decode.json": "eval_decode.json", "gpu_tests_full.log": "gpu_tests_full.log",
         "bench_force_dist_rows16.json": "bench_force_dist_rows16.json", "bench_force_dist_rows10.json": "bench_force_dist_rows10.json",
         "bench_n1_rows10.json": "bench_n1_rows10.json", "decode_attention_kernels.txt": "decode_attention_kernels.txt",
         "slice_defaults_check.txt": "slice_defaults_check.txt", "gemm_vs_vendor_blas.txt": "gemm_vs_vendor_blas.txt",
         "row_split_probe.txt": "row_split_probe.txt", "grid_barrier.txt": "grid_barrier.txt",
         "kernel_trace_steps.csv.gz": "kernel_trace_steps.csv.gz", "launch_floor.txt": "launch_floor.txt",
         "attention_study.txt": "attention_study.txt", "ring_depth_ab.txt": "ring_depth_ab.txt",
         "bench_launch_paths.txt": "bench_launch_paths.txt", "bench_force_dist_legs.json": "bench_force_dist_legs.json",
         "cold_operands.txt": "cold_operands.txt", "step_ab.txt": "step_ab.txt", "bench_wall.txt": "bench_wall.txt",
         "fused_update_bench.txt": "fused_update_bench.txt", "overlap_stats.txt": "overlap_stats.txt", "rccl_soak.txt": "rccl_soak.txt",
         "timeline_dist.txt": "timeline_dist.txt", "bench_repeat.txt": "bench_repeat.txt"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, b)))
        print("profiles/%s_%s" % (tag, b))
extra = {"r2/repro_raw.log": "repro_gc_capture_raw.log", "r2/repro_guarded.log": "repro_gc_capture_guarded.log",
         "r2/repro_rc.log": "repro_gc_capture_rc.log"}
for a, b in (extra.items() if "--with-repro" in sys.argv else []):
    p = os.path.join(root, "gpurun_out", a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, b)))
        print("profiles/%s_%s" % (tag, b))
