#!/bin/bash
# the whole GPU suite (C-level stderr uncaptured), smoke, the default bench line.  usage: tools/r3_full.sh <tag>
tag=${1:-r3g}
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --capture=sys --durations=12 > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 22 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - $tag <<PY
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json"))
print("value", round(d["value"]), round(d["ms_per_step"]*1e3,1), "us; roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), "; steady", round(d["steady_leg"]["value"]), "fuse", round(d["fuse_leg"]["value"]), "hd", round(d["hd_leg"]["value"]), {k:round(v["frac"],3) for k,v in d["hd_leg"]["roofline_by_kernel"].items()})
for k,v in d["reference_path_leg"].items():
    if isinstance(v, dict): print(" ", k, round(v["value"]), "fps", round(v["ms_per_step"]*1e3,1), "us")
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"])
PY
