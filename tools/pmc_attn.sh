#!/bin/bash
# What bounds the attention kernels?  SQ counters of the text self-attention shape (16 x 12 heads x 256 x 256 x 64), separate passes.
export TMPDIR=/tmp; out=gpurun_out/pmc_attn; rm -rf $out; mkdir -p $out
python3 tools/attn_probe.py 16 12 256 256 64 0 0.1 20
python3 tools/attn_probe.py 16 12 256 256 64 0 0.0 20
python3 tools/attn_probe.py 16 12 25 293 64 0 0.1 20
python3 tools/attn_probe.py 16 8 256 37 128 0 0.1 20
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/attn_probe.py 16 12 256 256 64 0 0.1 5 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob('gpurun_out/pmc_attn/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'fwd' if 'attn_fwd' in n else 'bwd' if 'attn_bwd' in n else None
        if k:
            a = agg[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, v in agg.items():
    d = {c: x[1] / x[0] for c, x in v.items()}
    print(k, {c: round(x) for c, x in sorted(d.items())})
    wc = d.get('SQ_WAVE_CYCLES')
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
            if c in d: print('   %-22s %.2f of wave cycles' % (c, d[c] / wc))
PY
