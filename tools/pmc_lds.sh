#!/bin/bash
# LDS bank conflicts / LDS activity of the 256-tile GEMM per operand layout (rocprofv3 PMC)
export TMPDIR=/tmp; mkdir -p gpurun_out/pmc2
for shape in "nt 4096 3072 768" "nn 4096 3072 768" "tn 18432 768 4688"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc2/$tag -- python3 tools/gemm_probe.py $shape 5 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc2/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    d = {k: v[1] / v[0] for k, v in agg.items()}
    print(f.split('/')[2], {k: round(v) for k, v in d.items()}, 'conflict/active = %.3f' % (d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
