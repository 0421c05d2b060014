#!/bin/bash
# experiment: k_surfel_pass grid = resident workgroups (SM_PASS_WG_PER_CU), configs[2] and KITTI 100 frames
for w in 0 5 6 7 8; do
  if [ $w -eq 0 ]; then unset SM_PASS_WG_PER_CU; else export SM_PASS_WG_PER_CU=$w; fi
  timeout -k 10 200 python bench.py --workload hd20m --steps 40 --warmup 5 --only-headline --no-cpu > gpurun_out/r3b_hd_wg$w.json 2> gpurun_out/r3b_hd_wg$w.err || { tail -n 5 gpurun_out/r3b_hd_wg$w.err; exit 1; }
  timeout -k 10 100 python bench.py --steps 100 --warmup 10 --only-headline --no-cpu > gpurun_out/r3b_k100_wg$w.json 2> gpurun_out/r3b_k100_wg$w.err || { tail -n 5 gpurun_out/r3b_k100_wg$w.err; exit 1; }
  python - $w <<PY
import json,sys
w=sys.argv[1]
for f in ("hd","k100"):
    d=json.load(open(f"gpurun_out/r3b_{f}_wg{w}.json"))
    k=d["kernels"]
    print("wg/cu",w,f,round(d["value"]),"fps",round(d["ms_per_step"]*1e3,1),"us | pass",round(k["k_surfel_pass"]["ms"]*1e3,1),"assoc_prep",round(k["k_assoc_prep"]["ms"]*1e3,1),"frac",round(d["roofline"]["frac"],3))
PY
done
