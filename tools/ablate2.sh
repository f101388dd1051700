#!/bin/bash
for a in 0 1 2 5; do echo "ABLATE=$a (0 = real kernel, 1 = no DMA in loop, 2 = no LDS reads/MFMA, 5 = ping-pong schedule)"
for s in "nt 4096 3072 6144" "nn 4096 3072 6144" "tn 4096 3072 6144" "tt 4096 3072 6144"; do GSTVD_GEMM_ABLATE=$a python3 tools/gemm_probe.py $s 20 2>&1 | grep TFLOP; done; done
