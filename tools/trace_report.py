#!/usr/bin/env python3
"""Report on one replayed train step from a rocprofv3 kernel trace (tools/trace_step.sh): per-queue span / busy time, phase
boundaries, time by number of kernels in flight, every interval with NO kernel running named by the kernels around it,
the memcpy / memset nodes of the graph, and the per-kernel totals.   usage: trace_report.py TRACE.csv[.gz] [min_gap_us=8] [step index]"""
import collections, csv, gzip, io, re, sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
fh = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rng_advance" in r["Kernel_Name"]]
# Steady-state replays only: the first steps of the trace are eager warm-up / capture, and the LAST complete step is the end of
# the timed burst (nothing queued behind it: round 2's timeline reported that one, with 1.2 ms of idle time and a 0.3 ms tail
# that the back-to-back replays in front of it do not have).  Report the median-length step among the ones that are followed by
# another replay, and list all of them.
spans = []
for k in range(len(idx) - 1):
    st = rows[idx[k]:idx[k + 1]]
    spans.append((max(int(r["End_Timestamp"]) for r in st) - int(st[0]["Start_Timestamp"]), k))
steady = [(sp, k) for sp, k in spans[:-1] if sp < 1.5 * min(s_ for s_, _ in spans)]
print("replayed steps followed by another replay: " + "  ".join("#%d %.3f ms (next starts +%.3f)" % (
    k, sp / 1e6, (int(rows[idx[k + 1]]["Start_Timestamp"]) - int(rows[idx[k]]["Start_Timestamp"])) / 1e6) for sp, k in steady))
pick = sorted(steady)[len(steady) // 2][1] if steady else len(idx) - 2
if len(sys.argv) > 3:
    pick = int(sys.argv[3])
a, b = idx[pick], idx[pick + 1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)


def short(n, w=56):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:w]


def ms(t):
    return (t - t0) / 1e6


print("steps in trace %d; reported step: %.3f ms, %d kernels (rng_advance -> next rng_advance %.3f ms)"
      % (len(idx), (t1 - t0) / 1e6, len(step), (int(rows[b]["Start_Timestamp"]) - t0) / 1e6))
by = collections.defaultdict(list)
for r in step:
    by[r["Queue_Id"]].append(r)
for q, v in sorted(by.items(), key=lambda kv: int(kv[1][0]["Start_Timestamp"])):
    s = int(v[0]["Start_Timestamp"]); e = max(int(r["End_Timestamp"]) for r in v)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
    print("  queue %-3s %4d kernels  first %.2f ms  last end %.2f ms  busy %.2f ms" % (q, len(v), ms(s), ms(e), busy / 1e6))
# phases
marks = []
for r in step:
    n = r["Kernel_Name"]
    gx = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    if "ce_fwd" in n or "ce_bwd" in n or "vl_split" in n or "adamw" in n or ("gemm_dma256_kernel" in n and gx == 1368) or "grouped" in n:
        marks.append("%s@%.2f-%.2f" % (short(n, 22), ms(int(r["Start_Timestamp"])), ms(int(r["End_Timestamp"]))))
print("  phase marks:", "  ".join(marks))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in step)
pts = sorted([(s, 1) for s, e, _ in ev] + [(e, -1) for s, e, _ in ev])
lvl, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
print("  time by kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
# idle intervals
cur_e, cur_r, idle, gaps = ev[0][1], ev[0][2], 0, []
for s, e, r in ev[1:]:
    if s > cur_e:
        idle += s - cur_e
        gaps.append((s - cur_e, cur_e, cur_r, r))
    if e > cur_e:
        cur_e, cur_r = e, r
nxt = int(rows[b]["Start_Timestamp"])
print("  no-kernel-running: %.3f ms inside the step in %d intervals (+ %.3f ms between its last kernel and the next step's first)"
      % (idle / 1e6, len(gaps), (nxt - t1) / 1e6))
hist = collections.Counter(min(int(g[0] / 1e3 // 2) * 2, 30) for g in gaps)
print("  idle-interval histogram (us: count):", " ".join("%d:%d" % (k, hist[k]) for k in sorted(hist)))
byphase = collections.Counter()
for g, at, prev, r in gaps:
    byphase[int(ms(at))] += g
print("  idle per ms of the step (us):", " ".join("%d:%.0f" % (k, v / 1e3) for k, v in sorted(byphase.items())))
print("  idle intervals >= %.0f us:" % min_gap)
for g, at, prev, r in gaps:
    if g / 1e3 >= min_gap:
        print("    %6.1f us at %.3f ms  after q%s %-40s before q%s %s" % (g / 1e3, ms(at), prev["Queue_Id"], short(prev["Kernel_Name"], 40),
                                                                      r["Queue_Id"], short(r["Kernel_Name"], 40)))
for r in step:
    n = r["Kernel_Name"]
    if "copyBuffer" in n or "fillBuffer" in n:
        print("  graph memcpy/memset node: %s q%s at %.3f ms (%.1f us)" % (short(n, 30), r["Queue_Id"], ms(int(r["Start_Timestamp"])),
                                                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = short(r["Kernel_Name"], 70)
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("  per-kernel totals (top 16):")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print("    %4d x %7.1f us = %7.3f ms  %s" % (n, t / n / 1e3, t / 1e6, k))
