#!/usr/bin/env python3
"""Summarise a rocprofv3 output tree (as produced by the commands in profiles/README.md):

  trace/      --kernel-trace --stats          -> per-kernel average duration
  pmc_fetch/  --pmc FETCH_SIZE                -> HBM bytes read per launch
  pmc_write/  --pmc WRITE_SIZE                -> HBM bytes written per launch

`--last K` restricts every average to the last K launches of each kernel, i.e. the launches of
bench.py's timed region (the first launches belong to its warm-up frames).  gfx950 correction
(/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE (KB) reports exactly 1/2 of the bytes
of a wide coalesced streaming read -> doubled; WRITE_SIZE (KB) is exact.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=None, help="recorded in the json (bench.py --steps of the profiled command)")
    ap.add_argument("--warmup", type=int, default=None)
    a = ap.parse_args()
    res = collections.defaultdict(dict)
    tr = glob.glob(os.path.join(a.dir, "trace", "*", "*_kernel_trace.csv"))
    if tr:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in per.items():
            v.sort()
            d = [x[1] for x in v][-a.last:] if a.last else [x[1] for x in v]
            res[k].update(calls=len(v), averaged=len(d), avg_us=sum(d) / len(d) / 1e3, min_us=min(d) / 1e3, max_us=max(d) / 1e3)
    for sub, ctr, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
        f = glob.glob(os.path.join(a.dir, sub, "*", "*_counter_collection.csv"))
        if not f:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == ctr:
                per[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in per.items():
            v.sort()
            d = [x[1] for x in v][-a.last:] if a.last else [x[1] for x in v]
            res[k][ctr + "_bytes_per_launch"] = sum(d) / len(d) * 1024.0 * mult
    for k, v in res.items():
        if "FETCH_SIZE_bytes_per_launch" in v and "WRITE_SIZE_bytes_per_launch" in v:
            v["hbm_bytes_per_launch"] = v["FETCH_SIZE_bytes_per_launch"] + v["WRITE_SIZE_bytes_per_launch"]
    rows = sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0))
    print(f"{'kernel':28s} {'calls':>6s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'HBM MB/launch':>14s}")
    for k, v in rows:
        if not k.startswith("k_"):
            continue
        hb = v.get("hbm_bytes_per_launch")
        print(f"{k:28s} {v.get('calls', 0):6d} {v.get('avg_us', 0):9.2f} {v.get('min_us', 0):8.2f} {v.get('max_us', 0):8.2f} "
              f"{(hb / 1e6 if hb else float('nan')):14.2f}")
    if a.out:
        json.dump({"last": a.last, "steps": a.steps, "warmup": a.warmup, "kernels": {k: v for k, v in rows if k.startswith('k_')}}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
