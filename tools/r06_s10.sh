#!/bin/bash
# Round 6, GPU session 10: two probes that price what is left of the step's structure with kernels that exist.
#  1. tools/ln_deferred_probe.py: an upper bound on what deferred normalisation can take off the 4096-row chain per LayerNorm site;
#  2. tools/graph_edge_probe.py: what a cross-stream dependency costs the source chain in a replayed graph (the 24 idle intervals of the step).
export TMPDIR=/tmp; out=gpurun_out/r06_s10; rm -rf $out; mkdir -p $out
timeout 300 python3 tools/ln_deferred_probe.py > $out/ln_deferred_probe.txt 2> $out/ln_deferred_probe.err; echo "rc $?"
cat $out/ln_deferred_probe.txt | cut -c1-300; tail -3 $out/ln_deferred_probe.err
timeout 300 python3 tools/graph_edge_probe.py > $out/graph_edge_probe.txt 2> $out/graph_edge_probe.err; echo "rc $?"
cat $out/graph_edge_probe.txt; tail -3 $out/graph_edge_probe.err
