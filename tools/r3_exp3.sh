#!/bin/bash
tag=${1:-r3d}
export SM_PASS_EPT=1
for w in 0 5 6 7; do
  if [ $w -eq 0 ]; then unset SM_PASS_WG_PER_CU; else export SM_PASS_WG_PER_CU=$w; fi
  timeout -k 10 200 python bench.py --workload hd20m --steps 40 --warmup 5 --only-headline --no-cpu > gpurun_out/${tag}_hd_wg$w.json 2> gpurun_out/${tag}_hd_wg$w.err || { tail -n 5 gpurun_out/${tag}_hd_wg$w.err; exit 1; }
  python - $tag $w <<PY
import json,sys
tag,w=sys.argv[1:3]
d=json.load(open(f"gpurun_out/{tag}_hd_wg{w}.json")); k=d["kernels"]
print("EPT1 wg/cu",w,"hd",round(d["value"]),"fps",round(d["ms_per_step"]*1e3,1),"us | pass",round(k["k_surfel_pass"]["ms"]*1e3,1),"assoc_prep",round(k["k_assoc_prep"]["ms"]*1e3,1),"frac",round(d["roofline"]["frac"],3))
PY
done
unset SM_PASS_WG_PER_CU
for cfg in "20 5" "100 10"; do set -- $cfg
  timeout -k 10 100 python bench.py --steps $1 --warmup $2 --only-headline --no-cpu > gpurun_out/${tag}_k$1.json 2> gpurun_out/${tag}_k$1.err || { tail -n 5 gpurun_out/${tag}_k$1.err; exit 1; }
  python - $tag $1 <<PY
import json,sys
tag,n=sys.argv[1:3]
d=json.load(open(f"gpurun_out/{tag}_k{n}.json")); k=d["kernels"]
print("EPT1 kitti",n,round(d["value"]),"fps",round(d["ms_per_step"]*1e3,1),"us | pass",round(k["k_surfel_pass"]["ms"]*1e3,1),"assoc_prep",round(k["k_assoc_prep"]["ms"]*1e3,1),"fixup",round(k["k_pass_fixup"]["ms"]*1e3,1))
PY
done
python tools/pass_trace_hd.py 6 2>&1 | tail -n 9
