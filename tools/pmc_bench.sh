#!/bin/bash
# HBM traffic of the dominant kernels inside the real train step (rocprofv3 PMC, one counter family per pass).
export TMPDIR=/tmp; mkdir -p gpurun_out/pmc3
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc3/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval-decode --no-breakdown --no-fp32 --no-h2d --graph off > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob('gpurun_out/pmc3/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        key = ('gemm_grouped_adamw_256' if 'grouped_adamw' in n else 'gemm_grouped_wgrad_256' if 'grouped' in n else 'gemm_pc256' if 'pc256' in n else 'gemm_dma256' if 'dma256' in n else 'gemm_dma128' if 'Li128ELi128' in n
               else 'gemm_dma64' if 'Li64ELi64' in n else 'adamw' if 'adamw' in n else 'ln_bwd' if 'ln_bwd' in n else 'ln_fwd' if 'ln_fwd' in n
               else 'attn' if 'attn' in n else None)
        if key:
            a = agg[key][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
out = {}
for k, v in agg.items():
    d = {c: x[1] / x[0] for c, x in v.items()}
    d['launches'] = max(x[0] for x in v.values())
    # gfx950: FETCH_SIZE counts 128-B requests as 64 B -> double it (MI355X_MICROARCH.md, HBM section); KB units
    d['hbm_bytes_per_launch'] = (2.0 * d.get('FETCH_SIZE', 0.0) + d.get('WRITE_SIZE', 0.0)) * 1024.0
    out[k] = d
json.dump(out, open('gpurun_out/pmc3/summary.json', 'w'), indent=1)
for k, d in out.items():
    print(k, {x: round(y, 1) for x, y in d.items()})
PY
