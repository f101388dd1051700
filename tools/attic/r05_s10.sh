#!/bin/bash
# Round 5, GPU session 10: does the operand sharing of a per-XCD queue decay over the rounds of a long launch?  (stand-alone, uniform problems)
export TMPDIR=/tmp; out=gpurun_out/r05_s10; rm -rf $out; mkdir -p $out
for mix in "72 0" "144 0"; do for ord in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/sa_${ord}_$c -- build/fused_update_bench $mix $ord > $out/sa.log 2>&1
  done
  python3 - "$mix" $ord <<'PY' | tee -a gpurun_out/r05_s10/rounds.txt
import csv, glob, collections, sys
mix, ord_ = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/r05_s10/sa_%s_*/*/*counter_collection.csv' % ord_):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'][:36]
        if 'grouped' in n:
            a = agg[n][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
nb = int(mix.split()[0])
for n, v in agg.items():
    d = {c: x[1] / x[0] for c, x in v.items()}
    print('stand-alone %s order %s  %-38s fetch %.2f GB (x2 corrected; operands once %.2f GB)  write %.2f GB' % (mix, ord_, n, 2 * d.get('FETCH_SIZE', 0) * 1024 / 1e9, nb * 31.5e6 / 1e9, d.get('WRITE_SIZE', 0) * 1024 / 1e9))
PY
  rm -rf $out/sa_${ord}_FETCH_SIZE $out/sa_${ord}_WRITE_SIZE
  build/fused_update_bench $mix $ord 2>&1 | tail -3 | tee -a $out/rounds.txt
done; done
