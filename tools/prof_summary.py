#!/usr/bin/env python3
"""Summarise a rocprofv3 output tree (as produced by the commands in profiles/README.md):

  trace/      --kernel-trace --stats          -> per-kernel average duration
  pmc_fetch/  --pmc FETCH_SIZE                -> HBM bytes read per launch
  pmc_write/  --pmc WRITE_SIZE                -> HBM bytes written per launch

  pmc_sq/     --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY ... -> VALU issue utilisation

`--last K` restricts every average to the launches of the last K frames, i.e. of bench.py's timed region (the earlier
launches belong to its warm-up frames and to its un-instrumented pass): a frame starts with a `k_prep` or `k_assoc_prep`
launch, so everything dispatched from the K-th last of those on is averaged -- kernels that run only on some frames
(`k_compact` on compacting frames, `k_cull_lazy` on the others) are averaged over their own launches in that window.  gfx950 correction
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


# FETCH_SIZE / WRITE_SIZE -> bytes on the memory fabric.  tools/calib_fetch.hip measured (profiles/fetch_calibration.json,
# MI355X, ROCm 7.2): FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming read at 4, 8 AND 16 bytes per
# lane (factor 2.000 each), and 64 B per random 8-byte gather (i.e., by the same 1/2 rule, one 128-byte line per gather);
# WRITE_SIZE is exact for coalesced stores at 4, 8 and 16 bytes per lane and counts 32 B per random 64-bit atomic.  So
# one factor per counter serves every kernel here -- the load width does not matter -- and a kernel that gathers shows
# its line over-fetch as real traffic, which is what the comparison with the algorithmic bytes is for.
def calibrate(cdir, out):
    """--calibrate: factors from a tools/calib_fetch run profiled with --pmc FETCH_SIZE (cdir/fetch) and WRITE_SIZE (cdir/write)."""
    known = {"k_calib_r4": 2 ** 30, "k_calib_r8": 2 ** 30, "k_calib_r16": 2 ** 30, "k_calib_g8": 8 * 2 ** 27,
             "k_calib_w4": 2 ** 30, "k_calib_w8": 2 ** 30, "k_calib_w16": 2 ** 30, "k_calib_a8": 8 * 2 ** 26}
    res = {"known_bytes": known, "fetch": {}, "write": {}, "note": "factor = bytes the kernel moves / (counter x 1024); a8 / g8: per byte of lane data"}
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = glob.glob(os.path.join(cdir, sub, "**", "*_counter_collection.csv"), recursive=True)
        if not f:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == ctr:
                m = re.search(r"(k_calib_[a-z0-9]+)", r["Kernel_Name"])
                if m:
                    per[m.group(1)].append(float(r["Counter_Value"]) * 1024.0)
        for k, v in sorted(per.items()):
            rep = sum(v) / len(v)
            res[sub][k.replace("k_calib_", "")] = {"reported_bytes": rep, "factor": (known[k] / rep) if rep > 0 else None, "launches": len(v),
                                                   "spread": (max(v) - min(v)) / rep if rep > 0 else None}
    json.dump(res, open(out, "w"), indent=1)
    for sub in ("fetch", "write"):
        for k, v in res[sub].items():
            print(f"{sub:6s} {k:5s} reported {v['reported_bytes'] / 1e6:10.1f} MB  factor {v['factor']}")


def counter_factor(cal, kind, default):
    """mean of the calibrated factors of the coalesced shapes of this counter (fetch: r4 r8 r16, write: w4 w8 w16)"""
    if not cal:
        return default
    shapes = ("r4", "r8", "r16") if kind == "fetch" else ("w4", "w8", "w16")
    f = [(cal.get(kind, {}).get(x) or {}).get("factor") for x in shapes]
    f = [x for x in f if x]
    return sum(f) / len(f) if f else default


def short(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--calibrate", action="store_true", help="dir holds fetch/ and write/ profiles of tools/calib_fetch: write the factor table to --out")
    ap.add_argument("--calibration", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "fetch_calibration.json"))
    ap.add_argument("--last", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=None, help="recorded in the json (bench.py --steps of the profiled command)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--compact-period", type=int, default=24, help="recorded in the json (bench.py --compact-period of the profiled command)")
    ap.add_argument("--simds", type=int, default=1024, help="SIMDs of the device (MI355X: 256 CUs x 4)")
    ap.add_argument("--mhz", type=float, default=2400.0, help="shader clock used to turn durations into cycles")
    a = ap.parse_args()
    if a.calibrate:
        return calibrate(a.dir, a.out or a.calibration)
    cal = json.load(open(a.calibration)) if os.path.exists(a.calibration) else None
    res = collections.defaultdict(dict)

    def window(per):
        """per: kernel -> sorted [(order key, value)]; returns kernel -> values inside the last `--last` frames"""
        # a frame starts with its preparation launch: k_prep, or k_assoc_prep when it also carries the previous frame's association
        starts = sorted(x[0] for name in ("k_prep", "k_assoc_prep") for x in per.get(name, []))
        if not a.last or len(starts) < a.last:
            return {k: [x[1] for x in v] for k, v in per.items()}
        t0 = starts[-a.last]
        return {k: [x[1] for x in v if x[0] >= t0] for k, v in per.items()}

    tr = glob.glob(os.path.join(a.dir, "trace", "*", "*_kernel_trace.csv"))
    if tr:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for v in per.values():
            v.sort()
        for k, d in window(per).items():
            if d:
                res[k].update(calls=len(per[k]), averaged=len(d), avg_us=sum(d) / len(d) / 1e3, min_us=min(d) / 1e3, max_us=max(d) / 1e3)
    for sub, ctr, mult, kind in (("pmc_fetch", "FETCH_SIZE", 2.0, "fetch"), ("pmc_write", "WRITE_SIZE", 1.0, "write")):
        f = glob.glob(os.path.join(a.dir, sub, "*", "*_counter_collection.csv"))
        if not f:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == ctr:
                per[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for v in per.values():
            v.sort()
        for k, d in window(per).items():
            if d:
                fac = counter_factor(cal, kind, mult)
                res[k][ctr + "_bytes_per_launch"] = sum(d) / len(d) * 1024.0 * fac
                res[k][ctr + "_factor"] = fac
    f = glob.glob(os.path.join(a.dir, "pmc_sq", "*", "*_counter_collection.csv"))
    if f:
        # SQ counters are quad-cycles (MI355X_MICROARCH.md): a wave64 VALU instruction occupies its SIMD for 4 cycles,
        # so  4 * SQ_ACTIVE_INST_VALU / (SIMDs * kernel cycles)  is the fraction of all VALU issue slots in use
        ctrs = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            ctrs[r["Counter_Name"]][short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for name, per in ctrs.items():
            for v in per.values():
                v.sort()
            for k, d in window(per).items():
                if d:
                    res[k][name] = sum(d) / len(d)
        for k, v in res.items():
            if "SQ_ACTIVE_INST_VALU" in v and v.get("avg_us"):
                v["valu_issue_util"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (a.simds * v["avg_us"] * a.mhz)
            if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
                v["wave_wait_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    for k, v in res.items():
        if "FETCH_SIZE_bytes_per_launch" in v and "WRITE_SIZE_bytes_per_launch" in v:
            v["hbm_bytes_per_launch"] = v["FETCH_SIZE_bytes_per_launch"] + v["WRITE_SIZE_bytes_per_launch"]
    rows = sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0))
    print(f"{'kernel':28s} {'calls':>6s} {'avgd':>5s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'HBM MB/launch':>14s} {'VALU util':>10s} {'wave wait':>10s}")
    for k, v in rows:
        if not k.startswith("k_"):
            continue
        hb = v.get("hbm_bytes_per_launch")
        print(f"{k:28s} {v.get('calls', 0):6d} {v.get('averaged', 0):5d} {v.get('avg_us', 0):9.2f} {v.get('min_us', 0):8.2f} {v.get('max_us', 0):8.2f} "
              f"{(hb / 1e6 if hb else float('nan')):14.2f} {v.get('valu_issue_util', float('nan')):10.2f} {v.get('wave_wait_frac', float('nan')):10.2f}")
    if a.out:
        json.dump({"last": a.last, "steps": a.steps, "warmup": a.warmup, "compact_period": a.compact_period, "kernels": {k: v for k, v in rows if k.startswith('k_')}}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
