#!/bin/bash
# Round 5, GPU session 21: the final grouped weight-gradient launch started BEFORE the text embedding's backward and the last column
# reductions (GSTVD_EARLY_WGRAD 0 / 1): model / pipeline / fused-update tests, step A/B, tail of the timeline.
export TMPDIR=/tmp; out=gpurun_out/r05_s21; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "fused or pipeline or model or full_config or round3 or round4 or round5 or dp" 2>&1 | tail -3 | tee $out/tests.log
bash tools/r04_step_ab.sh GSTVD_EARLY_WGRAD 0 1 2>&1 | tee $out/step_ab.txt
bash tools/r04_step_ab.sh GSTVD_EARLY_WGRAD 0 1 2>&1 | tee -a $out/step_ab.txt
