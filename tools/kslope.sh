#!/bin/bash
for lay in nt nn; do for K in 256 768 1536 3072 6144; do python3 tools/gemm_probe.py $lay 4096 3072 $K 30 2>&1 | grep TFLOP; done; done
for K in 256 768 1536 3072; do python3 tools/gemm_probe.py nt 4096 768 $K 30 2>&1 | grep TFLOP; done
