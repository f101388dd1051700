#!/bin/bash
# Round 5, GPU session 12: why does the full-size bit-identity test fail with the L2 prefetch build?
export TMPDIR=/tmp; out=gpurun_out/r05_s12; rm -rf $out; mkdir -p $out
L=gst_visdial_amd/lib
for v in 1 0 1; do
  cp $L/libgstvd_hip_pf$v.so $L/libgstvd_hip.so
  GSTVD_TEST_CHILD=1 timeout 600 python -m pytest "tests/test_fused_update_gpu.py::test_full_size_step_with_the_fused_update_is_bit_identical_to_the_two_launches" -x -q 2>&1 | grep -E "^E |passed|failed" | cut -c1-400 | head -20 | tee -a $out/bitid_pf$v.log
done
cp $L/libgstvd_hip_pf0.so $L/libgstvd_hip.so
