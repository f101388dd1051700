#!/bin/bash
for a in 0 3 5; do echo "VARIANT=$a"
for s in "nt 4096 3072 768" "nn 4096 3072 768" "tn 18432 768 4688" "nt 4096 3072 6144"; do GSTVD_GEMM_ABLATE=$a python3 tools/gemm_probe.py $s 30 2>&1 | grep TFLOP; done; done
