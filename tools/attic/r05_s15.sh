export TMPDIR=/tmp; out=gpurun_out/r05_s15; rm -rf $out; mkdir -p $out
for i in 1 2 3; do
GSTVD_TEST_CHILD=1 timeout 600 python -m pytest "tests/test_fused_update_gpu.py::test_full_size_step_with_the_fused_update_is_bit_identical_to_the_two_launches" -x -q 2>&1 | grep -E "^E |passed|failed|Error" | cut -c1-300 | head -12 | tee -a $out/bitid.log
done
