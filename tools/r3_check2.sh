#!/bin/bash
tag=${1:-r3e}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py tests/test_deferred_compaction.py tests/test_shard_stream.py tests/test_kat.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-hd-leg --no-cpu > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; rc=$?
echo "bench rc=$rc"; tail -n 3 gpurun_out/${tag}_bench.err
python - $tag <<PY
import json,sys
d=json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json"))
print("value", round(d["value"]), round(d["ms_per_step"]*1e3,1), "us; steady", round(d["steady_leg"]["value"]), "fuse", round(d["fuse_leg"]["value"]))
for k,v in d["reference_path_leg"].items():
    if isinstance(v, dict): print(" ", k, round(v["value"]), "fps", round(v["ms_per_step"]*1e3,1), "us")
    else: print(" ", k, v)
PY
