#!/bin/bash
# Quick A/B step on the GPU box: the parity subset (fuzz, deferred compaction, KATs), then the three bench lines with their
# dominant kernels.  usage (through gpurun): bash tools/ab.sh <tag>   -> gpurun_out/<tag>_{k20,k100,hd}.json
set -e
tag=$1; shift
python -m pytest tests/test_fuzz_gpu.py tests/test_deferred_compaction.py tests/test_kat.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1 || { tail -n 30 gpurun_out/${tag}_pytest.log; exit 1; }
tail -n 1 gpurun_out/${tag}_pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu --no-steady-leg --no-reference-path-leg --no-hd-leg > gpurun_out/${tag}_k20.json 2> gpurun_out/${tag}.err
python bench.py --steps 100 --warmup 10 --no-cpu --only-headline > gpurun_out/${tag}_k100.json 2>> gpurun_out/${tag}.err
python bench.py --workload hd20m --steps 40 --warmup 5 --no-cpu > gpurun_out/${tag}_hd.json 2>> gpurun_out/${tag}.err
for f in k20 k100 hd; do python - $tag $f <<PY
import json,sys
d=json.load(open("gpurun_out/%s_%s.json"%(sys.argv[1],sys.argv[2])))
fl=d.get("fuse_leg") or {}
print(sys.argv[2], round(d["value"]), round(d["ms_per_step"]*1e3,2), {k:round(v["ms"]*1e3,2) for k,v in d["kernels"].items() if k in ("k_assoc_prep","k_surfel_pass","k_pass_fixup")}, "fuse", fl.get("value") and round(fl["value"]), {k:round(v["ms"]*1e3,2) for k,v in (fl.get("kernels") or {}).items() if k in ("k_assoc_prep","k_surfel_pass")})
PY
done
