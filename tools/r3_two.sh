#!/bin/bash
# two-launch frame: the GPU suite with it on (default), then A/B bench lines.  usage: tools/r3_two.sh <tag>
tag=${1:-r3two}
timeout -k 10 900 python -m pytest tests -m gpu -x -q --capture=sys --durations=8 > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 16 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit 1
for tw in 0 1; do for cfg in "20 5" "100 10"; do set -- $cfg
  SM_TWO_LAUNCH=$tw timeout -k 10 300 python bench.py --steps $1 --warmup $2 --only-headline --no-cpu-baseline > gpurun_out/${tag}_b_${tw}_$1.json 2>> gpurun_out/${tag}_bench.err || exit 1
  python - gpurun_out/${tag}_b_${tw}_$1.json $tw <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d.get("kernels_us") or d.get("kernel_table") or {}
print("two", sys.argv[2], "steps", d["steps"], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {a:(round(b,1) if isinstance(b,(int,float)) else b) for a,b in list(k.items())[:8]} if isinstance(k,dict) else "")
PY
done; done
SM_TWO_LAUNCH=1 timeout -k 10 300 python bench.py --workload hd20m --steps 40 --warmup 5 --only-headline --no-cpu-baseline > gpurun_out/${tag}_hd_1.json 2>> gpurun_out/${tag}_bench.err && SM_TWO_LAUNCH=0 timeout -k 10 300 python bench.py --workload hd20m --steps 40 --warmup 5 --only-headline --no-cpu-baseline > gpurun_out/${tag}_hd_0.json 2>> gpurun_out/${tag}_bench.err
python - <<PY
import json
for t in (0,1):
    d=json.load(open("gpurun_out/${tag}_hd_%d.json"%t)); print("hd two",t,round(d["value"]),round(d["ms_per_step"]*1e3,1),"us")
PY
