#!/bin/bash
# quick check of a kernel change: parity subset, headline at (20,5) and (100,10), the 23-frame launch trace.  usage: tools/r3_quick.sh <tag>
tag=${1:-r3q}
timeout -k 10 600 python -m pytest tests/test_kat.py tests/test_gpu_parity.py tests/test_deferred_compaction.py tests/test_fuzz_gpu.py tests/test_shard_stream.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log; [ $rc -ne 0 ] && exit 1
for cfg in "20 5" "100 10"; do set -- $cfg
  timeout -k 10 300 python bench.py --steps $1 --warmup $2 --only-headline --no-cpu-baseline > gpurun_out/${tag}_$1.json 2>> gpurun_out/${tag}.err || exit 1
  python - gpurun_out/${tag}_$1.json <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']
print("steps", d["steps"], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {n:round(v['ms']*1e3,1) for n,v in k.items() if n in('k_assoc_prep','k_surfel_pass')})
PY
done
python tools/pass_trace.py 23 > gpurun_out/${tag}_pt23.txt 2>&1; grep -A8 "k_assoc_prep:" gpurun_out/${tag}_pt23.txt
