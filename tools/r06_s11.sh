#!/bin/bash
# Round 6, GPU session 11: does any switch of the HIP runtime change what a cross-stream edge costs inside a replayed graph
# (tools/graph_edge_probe.py: +8..25 us per edge on the SOURCE chain)?  One process per setting, each under its own timeout.
export TMPDIR=/tmp; out=gpurun_out/r06_s11; rm -rf $out; mkdir -p $out
run() {
  tag=$1; shift
  echo "=== $tag" | tee -a $out/edge_env_sweep.txt
  env "$@" timeout 120 python3 tools/graph_edge_probe.py 2>$out/err_$tag.txt | tee -a $out/edge_env_sweep.txt
  echo "rc ${PIPESTATUS[0]}" | tee -a $out/edge_env_sweep.txt
}
run default X=1
run opt_flush0 AMD_OPT_FLUSH=0
run opt_flush1 AMD_OPT_FLUSH=1
run opt_flush3 AMD_OPT_FLUSH=3
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run skip_release DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graphq1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run graphq2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run cpwait1 GPU_STREAMOPS_CP_WAIT=1
run cpwait0 GPU_STREAMOPS_CP_WAIT=0
run dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run dynq1 DEBUG_HIP_DYNAMIC_QUEUES=1
