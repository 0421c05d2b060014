#!/bin/bash
# A/B lines after the one-tile-per-thread flag workgroups.  usage: tools/r3_ab2.sh <tag>
tag=${1:-r3ab2}
timeout -k 10 600 python -m pytest tests/test_kat.py tests/test_gpu_parity.py tests/test_deferred_compaction.py tests/test_fuzz_gpu.py tests/test_shard_stream.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/${tag}_pytest.log; [ $rc -ne 0 ] && exit 1
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $BARGS --only-headline --no-cpu-baseline > gpurun_out/${tag}_$name.json 2>> gpurun_out/${tag}_bench.err || return 1
  python - gpurun_out/${tag}_$name.json $name <<PY
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']
print(sys.argv[2], "value", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us", {n:round(v['ms']*1e3,1) for n,v in k.items() if n in('k_assoc_prep','k_surfel_pass','k_pass_fixup')})
PY
}
for cfg in "20 5" "100 10"; do set -- $cfg; BARGS="--steps $1 --warmup $2"
  run def_$1 X=1 && run s1_$1 SM_PASS_SPLIT=1 || exit 1
done
BARGS="--workload hd20m --steps 40 --warmup 5"; run hd X=1
python tools/pass_trace.py 110 > gpurun_out/${tag}_pt110.txt 2>&1
