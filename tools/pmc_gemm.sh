#!/bin/bash
# L2 hit rate / HBM fetch of the big GEMM shapes (rocprofv3 PMC, separate passes). Run on the GPU box from the repo root.
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
for shape in "nt 4096 3072 768" "tn 3072 768 4096" "nn 4096 768 3072"; do
  tag=$(echo $shape | tr ' ' '_')
  python3 tools/gemm_probe.py $shape 20
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc/$tag-l2 -- python3 tools/gemm_probe.py $shape 5 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/$tag-fetch -- python3 tools/gemm_probe.py $shape 5 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    print(f.split('/')[2], {k: round(v[1] / v[0], 1) for k, v in agg.items()})
PY
