#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of tools/gemm_bench.py: one line per distinct (kernel, grid) with its launch
configuration and resources -- how the vendor BLAS kernels (Cijk_* Tensile names: macro tile MT, depthU DU, wave tile,
LDS / direct-to-LDS / prefetch flags) differ from this library's gemm_* kernels on the same shapes.
usage: vendor_kernel_config.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not (n.startswith("Cijk") or "gemm" in n.lower()):
        continue
    key = (n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    a = agg.setdefault(key, dict(n=0, t=0.0, tmin=1e30, r=r))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a["n"] += 1
    a["t"] += d
    a["tmin"] = min(a["tmin"], d)
print("calls  avg_us  min_us  grid(threads)  wg  workgroups  LDS_B  VGPR AGPR SGPR scratch  kernel")
for (n, gx, gy, gz, wx), a in agg.items():
    r = a["r"]
    wgs = (int(gx) // int(wx)) * int(gy) * int(gz)
    print("%5d %7.1f %7.1f  %sx%sx%s  %s  %d  %s  %s %s %s %s  %s" % (
        a["n"], a["t"] / a["n"], a["tmin"], gx, gy, gz, wx, wgs, r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"],
        r["SGPR_Count"], r["Scratch_Size"], n))
