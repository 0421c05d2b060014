#!/bin/bash
# compaction period sweep on the two-launch code.  usage: tools/r3_period.sh <tag>
tag=${1:-r3per}
for st in 100 200; do for p in 12 16 24 32 48; do
  timeout -k 10 300 python bench.py --steps $st --warmup 10 --compact-period $p --only-headline --no-cpu-baseline > gpurun_out/${tag}_${st}_$p.json 2>> gpurun_out/${tag}.err || exit 1
  python - gpurun_out/${tag}_${st}_$p.json $p <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']
print("period", sys.argv[2], "steps", d["steps"], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {n:(round(v['ms']*1e3,1), v['launches']) for n,v in k.items() if n in('k_assoc_prep','k_surfel_pass','k_compact','k_conflict')})
PY
done; done
