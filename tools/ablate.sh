#!/bin/bash
for a in 0 1 2; do echo "ABLATE=$a (0 = real kernel, 1 = no DMA in loop, 2 = no LDS reads/MFMA)"; GSTVD_GEMM_ABLATE=$a python3 tools/gemm_probe.py nt 4096 3072 768 30 2>&1 | grep TFLOP; GSTVD_GEMM_ABLATE=$a python3 tools/gemm_probe.py nt 4688 18432 768 20 2>&1 | grep TFLOP;  GSTVD_GEMM_ABLATE=$a python3 tools/gemm_probe.py nn 4096 3072 768 30 2>&1 | grep TFLOP; done
