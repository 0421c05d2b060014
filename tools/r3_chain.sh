#!/bin/bash
# depth-chain stage: parity (pre-processing tests, fuzz with the chain on, golden fixtures), then the reference-path leg.  usage: tools/r3_chain.sh <tag>
tag=${1:-r3chain}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py tests/test_golden.py tests/test_shard_stream.py tests/test_kat.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-fuse-leg --no-steady-leg --no-hd-leg --no-cpu-baseline > gpurun_out/${tag}_b.json 2> gpurun_out/${tag}.err || exit 1
python - gpurun_out/${tag}_b.json <<PY
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d["reference_path_leg"].items():
    if isinstance(v, dict): print(" ", k, round(v["value"]), "fps", round(v["ms_per_step"]*1e3,1), "us", {n:round(x['ms']*1e3,1) for n,x in (v.get('kernels') or {}).items() if n in ('k_assoc_prep','k_surfel_pass')})
PY
python tools/chain_trace.py > gpurun_out/${tag}_chain_trace.txt 2>&1; tail -n 12 gpurun_out/${tag}_chain_trace.txt
