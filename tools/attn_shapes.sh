#!/bin/bash
# forward / backward time of the step's six attention shapes (bf16, dropout 0.1), stand-alone
for sh in "16 12 256 256 64 0" "16 8 256 37 128 0" "16 8 37 256 128 0" "16 8 37 37 128 0" "16 12 25 25 64 1" "16 12 25 293 64 0"; do
  python3 tools/attn_probe.py $sh 0.1 50 | tail -1
done
