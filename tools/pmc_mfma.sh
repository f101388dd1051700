#!/bin/bash
# MFMA pipe utilisation of the isolated GEMM shapes of the step: SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x SIMDs).
# SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (16 cycles per v_mfma_f32_16x16x32_bf16: it equals
# 16 x M*N*K/8192 exactly); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_mfma
# the last six are the connection layer's (vilbert_dialog.py:646-773): QKV2 / text FFN up (= 4096x3072x768), text FFN down, biOutput
# dense2 (4096x768x1024) and its input gradient, QKV1 (592x3072x1024), biOutput dense1 / vision FFN (592x1024x1024)
for shape in "nt 4096 3072 768" "nn 4096 768 3072" "nt 4096 768 3072" "tn 18432 768 4688" "nt 4688 18432 768" \
             "nn 4096 3072 768" "nt 4096 768 1024" "nn 4096 1024 768" "nt 592 3072 1024" "nt 592 1024 1024" "nn 592 1024 1024"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma/$tag -- python3 tools/gemm_probe.py $shape 5 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/pmc_mfma/*/*/*counter_collection.csv')):
    tag = f.split('/')[2]
    agg = collections.defaultdict(lambda: [0, 0.0])
    dur = []
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    d = {k: v[1] / v[0] for k, v in agg.items()}
    us = sum(dur) / len(dur) / 1e3
    cyc = d.get('GRBM_GUI_ACTIVE', 0.0)
    util = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc / 8.0 * 1024) if cyc else None
    out[tag] = dict(us_under_pmc=round(us, 1), mfma_busy_cycles=d.get('SQ_VALU_MFMA_BUSY_CYCLES'), gui_active_cycles=cyc, mfma_pipe_util=util)
    print(tag, out[tag])
json.dump(out, open('gpurun_out/pmc_mfma/summary.json', 'w'), indent=1)
PY
