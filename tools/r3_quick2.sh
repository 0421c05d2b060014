#!/bin/bash
# parity subset + full-size configs + headline at three workloads.  usage: tools/r3_quick2.sh <tag>
tag=${1:-r3q}
timeout -k 10 900 python -m pytest tests/test_kat.py tests/test_gpu_parity.py tests/test_deferred_compaction.py tests/test_fuzz_gpu.py tests/test_shard_stream.py tests/test_configs_full_size.py tests/test_rig.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log; [ $rc -ne 0 ] && exit 1
for cfg in "20 5 kitti" "100 10 kitti" "40 5 hd20m"; do set -- $cfg
  timeout -k 10 300 python bench.py --steps $1 --warmup $2 --workload $3 --only-headline --no-cpu-baseline > gpurun_out/${tag}_$3_$1.json 2>> gpurun_out/${tag}.err || exit 1
  python - gpurun_out/${tag}_$3_$1.json <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']
print(d["config"]["workload"][:12], "steps", d["steps"], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {n:round(v['ms']*1e3,1) for n,v in k.items() if n in('k_assoc_prep','k_surfel_pass')})
PY
done
