#!/bin/bash
# where to switch k_surfel_pass from quarter-tile units to whole tiles + resident grid (tiles of the slot estimate).  usage: tools/r3_thr.sh <tag>
tag=${1:-r3thr}
for st in 100 200; do for thr in 4096 8192 16384 1000000; do
  SM_PASS_PERSIST_TILES=$thr timeout -k 10 300 python bench.py --steps $st --warmup 10 --only-headline --no-cpu-baseline > gpurun_out/${tag}_${st}_$thr.json 2>> gpurun_out/${tag}.err || exit 1
  python - gpurun_out/${tag}_${st}_$thr.json $thr <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']
print("threshold", sys.argv[2], "steps", d["steps"], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {n:round(v['ms']*1e3,1) for n,v in k.items() if n in('k_assoc_prep','k_surfel_pass')})
PY
done; done
