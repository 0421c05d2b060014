#!/usr/bin/env python3
"""Idle time between consecutive kernels of one rocprofv3 --kernel-trace CSV (GPU-side launch gaps).
usage: trace_gaps.py <dir with *_kernel_trace.csv> [--last N kernels]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
last = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "--last" else 0
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
if last:
    rows = rows[-last:]
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[f"{n0} -> {n1}"].append((s1 - e0) / 1e3)
tot_k = sum(e - s for s, e, _ in rows) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{len(rows)} kernels, span {span:.1f} us, in kernels {tot_k:.1f} us ({tot_k / span:.2%}), idle {span - tot_k:.1f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"  {k:48s} n={len(v):4d}  median {v2[len(v2)//2]:7.2f} us  mean {sum(v)/len(v):7.2f}  max {v2[-1]:8.2f}")
