#!/bin/bash
# tile sequences of k_surfel_pass<4> (SM_PASS_SEQ; workgroups = 4 x sequences).  usage: tools/r3_seq.sh <tag>
tag=${1:-r3seq}
for cfg in "20 5" "100 10"; do set -- $cfg; for q in 256 320 384 448 512; do
  SM_PASS_SEQ=$q timeout -k 10 300 python bench.py --steps $1 --warmup $2 --only-headline --no-cpu-baseline > gpurun_out/${tag}_$1_$q.json 2>> gpurun_out/${tag}.err || exit 1
  python - gpurun_out/${tag}_$1_$q.json $q <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']
print("seq", sys.argv[2], "steps", d["steps"], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {n:round(v['ms']*1e3,1) for n,v in k.items() if n in('k_assoc_prep','k_surfel_pass')})
PY
done; done
