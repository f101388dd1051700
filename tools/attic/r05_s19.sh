#!/bin/bash
# Round 5, GPU session 19: does the step's TABLE of weight-gradient problems reproduce the in-step operand refetch stand-alone?
# (plain grouped launch: fabric reads = operand fetches only) -- whole table / long-K part / uniform table, GSTVD_GROUP_ORDER 0 / 1 / 3.
export TMPDIR=/tmp; out=gpurun_out/r05_s19; rm -rf $out; mkdir -p $out
for which in all long uniform; do for v in 0 1 3; do
  export GSTVD_GROUP_ORDER=$v
  python3 tools/wgrad_table_probe.py $which 2>/dev/null | tail -1 | tee -a $out/table_probe.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc -- python3 tools/wgrad_table_probe.py $which > /dev/null 2>&1
  python3 - <<'PY' | tee -a gpurun_out/r05_s19/table_probe.txt
import csv, glob
n, tot = 0, 0.0
for f in glob.glob('gpurun_out/r05_s19/pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'grouped' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            n += 1; tot += float(r['Counter_Value'])
print('    FETCH_SIZE x 2 = %.3f GB per launch (%d launches)' % (2 * tot / max(n, 1) * 1024 / 1e9, n))
PY
  rm -rf $out/pmc
done; done
