#!/bin/bash
# L2-miss traffic of the batched weight-gradient GEMM of tools/overlap_bench.cpp (24 x [3072 x 768 x 4096]: 0.75 GB of operands if every
# operand tile were fetched once, 3.5 GB if every tile fetched its own) under different XCD chunk sizes.  usage: tools/pmc_overlap.sh
export TMPDIR=/tmp; out=gpurun_out/pmc_overlap; rm -rf $out; mkdir -p $out
for chs in 0 3 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GSTVD_GROUP_CHUNK_LOG2=$chs rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$chs/$c -- ${PMC_BIN:-build/overlap_bench 24 8} > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for chs in (0, 3, 6):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob('gpurun_out/pmc_overlap/%d/*/*/*counter_collection.csv' % chs):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'][:40]
            a = agg[n][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for n, v in agg.items():
        d = {c: x[1] / x[0] for c, x in v.items()}
        print('chunk 2^%d  %-42s launches %3d  fetch %.2f GB (x2 corrected)  write %.2f GB' % (chs, n, max(x[0] for x in v.values()), 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e9, d.get('WRITE_SIZE', 0) * 1024 / 1e9))
PY
