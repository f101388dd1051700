#!/usr/bin/env python3
"""Per-kernel durations and gaps of ONE token step inside the replayed decode graph, from a rocprofv3 kernel trace of
tools/decode_debug.py:   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dec_trace -- python3 tools/decode_debug.py
                         python3 tools/decode_timeline.py gpurun_out/dec_trace"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "") + " grid=" + r.get("Grid_Size_X", "?"))
        for r in csv.DictReader(open(f))]
rows.sort()
# the last call: find the last sample_topk kernels (18 per call)
idx = [i for i, r in enumerate(rows) if "sample_topk" in r[2]]
last = idx[-18:]
a, b = last[9], last[10]                      # one token step in the middle: between two consecutive sampling kernels
step = rows[a + 1:b + 1]
t0 = rows[a][1]
print("token step: %d kernels, %.1f us wall" % (len(step), (step[-1][1] - t0) / 1e3))
agg = collections.OrderedDict()
prev_end = t0
busy = gap = 0
for s, e, n in step:
    k = n.split("(")[0][:48] + n[n.rfind(" grid="):]
    d = agg.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(0, s - prev_end) / 1e3
    busy += e - s; gap += max(0, s - prev_end)
    prev_end = e
print("busy %.1f us, gaps %.1f us" % (busy / 1e3, gap / 1e3))
for k, (n, dur, g) in agg.items():
    print("%3d x %6.2f us  (+gap %5.2f us each)  %s" % (n, dur / n, g / n, k))
call = rows[idx[-18] - 1000 if idx[-18] > 1000 else 0:]
print("whole call (last 18 steps): %.2f ms" % ((rows[last[-1]][1] - rows[last[0]][0]) / 1e6))
