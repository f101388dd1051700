#!/bin/bash
# VGPRs / scratch / occupancy of every kernel of one source file of the library (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed),
# and a non-zero exit code when a kernel listed in tools/kernel_resources.expect exceeds its scratch budget.  Run after EVERY edit of a hot
# kernel: round 5 shipped a start barrier for two commits whose call cost gemm_pc256_grouped_adamw_kernel 256 B/lane of scratch (12 B
# without it) -- 0.38 GB of extra writes per launch that only showed up as a WRITE_SIZE nobody expected.
#   bash tools/kernel_resources.sh gemm_dma256.hip [attention.hip ...]        (default: every .hip under gst_visdial_amd/csrc)
cd "$(dirname "$0")/../gst_visdial_amd/csrc" || exit 2
files=${@:-$(ls *.hip)}
rc=0
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c $f -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - |
    awk -v f=$f '{n=$3; v=$5; s=$8; o=$11; printf "%-16s %-100s VGPRs %3s scratch %4s B occupancy %s\n", f, substr(n,1,100), v, s, o}' > /tmp/kr_$$.txt
  cat /tmp/kr_$$.txt
  while read -r pat budget; do
    [ -z "$pat" ] && continue
    case "$pat" in \#*) continue;; esac
    awk -v p="$pat" -v b="$budget" '$2 ~ p { if ($6 + 0 > b + 0) { printf "  !! %s: scratch %s B > budget %s B\n", $2, $6, b; bad = 1 } } END { exit bad }' /tmp/kr_$$.txt || rc=1
  done < ../../tools/kernel_resources.expect
  rm -f /tmp/kr_$$.o /tmp/kr_$$.txt
done
exit $rc
