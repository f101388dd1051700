#!/bin/bash
# VERDICT r2 #2 proposed running the decoder's 16 rows (M = 400) as two 8-row chains (M = 200) on two streams.  That shortens
# the decoder phase only if a launch of the chain gets faster with fewer rows.  This probe times every kernel class of a decoder
# layer back to back at full / half / quarter rows: GEMMs (M = 400 / 200 / 100), LayerNorm forward / backward, self- and
# cross-attention (batch 16 / 8 / 4).
echo "== GEMMs of a decoder layer by rows"
python3 tools/gemm_bench.py mscale 2>/dev/null | grep -v amdgpu
echo "== LayerNorm (400 x 768 and fewer rows)"
python3 tools/ln_probe.py mscale 2>/dev/null | grep "M="
echo "== attention: decoder self (25 x 25, causal) and cross (25 x 293) by batch rows"
for b in 16 8 4; do
  python3 tools/attn_probe.py $b 12 25 25 64 1 0.1 50 2>/dev/null | tail -1
  python3 tools/attn_probe.py $b 12 25 293 64 0 0.1 50 2>/dev/null | tail -1
done
