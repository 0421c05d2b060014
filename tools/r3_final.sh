#!/bin/bash
# end-of-round validation: the GPU suite, a fuzz soak, smoke, the default bench line, the 100-frame and configs[2] headline lines,
# the one-rank rehearsal of the N-rank line.  usage: tools/r3_final.sh <tag>
tag=${1:-r03}
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 10 gpurun_out/${tag}_pytest.log; [ $rc -ne 0 ] && exit 1
SM_FUZZ_SEEDS=800 timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > gpurun_out/${tag}_soak.log 2>&1; rc=$?
tail -n 2 gpurun_out/${tag}_soak.log; [ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout -k 10 400 python bench.py > gpurun_out/bench_${tag}_default.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --only-headline > gpurun_out/bench_${tag}_kitti_s100_w10.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 python bench.py --workload hd20m --steps 40 --warmup 5 --only-headline > gpurun_out/bench_${tag}_hd20m.json 2>> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 400 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > gpurun_out/bench_${tag}_ranks_world1.json 2>> gpurun_out/${tag}_bench.err || exit 1
python - $tag <<PY
import json,sys
t=sys.argv[1]
d=json.load(open(f"gpurun_out/bench_{t}_default.json"))
print("value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us; roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), "traffic", d["roofline"].get("traffic"))
print("steady", round(d["steady_leg"]["value"]), "fuse", round(d["fuse_leg"]["value"]), "hd", round(d["hd_leg"]["value"]), {k:round(v["frac"],3) for k,v in d["hd_leg"]["roofline_by_kernel"].items()}, "match", d["hd_leg"].get("final_counts_match_gpu"))
for k,v in d["reference_path_leg"].items():
    if isinstance(v, dict): print(" ", k, round(v["value"]), "fps", round(v["ms_per_step"]*1e3,1), "us")
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"])
r=json.load(open(f"gpurun_out/bench_{t}_ranks_world1.json"))
print("ranks", round(r["value"]), r["config"].get("multi_gpu"), r.get("rccl"), {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.get("sharded_leg",{}).items() if k in ("value","sharded_over_plain","ms_per_step")})
PY
