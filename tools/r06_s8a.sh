#!/bin/bash
# quick check of the XCD-local hand-off before the full session 8: op tests only, hard time limit
export TMPDIR=/tmp; out=gpurun_out/r06_s8; mkdir -p $out
timeout 300 python3 -X faulthandler -m pytest tests/test_round6_gpu.py -x -q -m gpu -p no:cacheprovider -k "layernorm_epilogue" > $out/quick_tests.log 2>&1; echo "rc=$?" >> $out/quick_tests.log
grep -n "Error\|error\|passed\|failed\|rc=\|assert" $out/quick_tests.log | head -20
