#!/bin/bash
# fused weight-gradient + AdamW launch: parity tests, then the whole-step A/B (GSTVD_FUSE_UPDATE 0 / 1) on one and on the default slice list
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests/test_fused_update_gpu.py "tests/test_round3_gpu.py::test_bench_path_graph_replay_with_pipeline_matches_oracle_in_train_mode" -x -q -p no:cacheprovider 2>&1 | tail -15
for rnd in 1 2; do
  for cl in 500 "22,27,27,27,27,192"; do
    for f in 0 1; do
      r=$(GSTVD_FUSE_UPDATE=$f python3 bench.py --steps 30 --warmup 5 --chunk-list $cl --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d 2>gpurun_out/r4/fuse_err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
      echo "round $rnd chunk-list $cl GSTVD_FUSE_UPDATE=$f: $r"
    done
  done
done
tail -3 gpurun_out/r4/fuse_err.txt
