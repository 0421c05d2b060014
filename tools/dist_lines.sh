#!/bin/bash
# The multi-GPU modes rehearsed with one rank on the GPU box (RCCL bound at world = 1): tests of the sharded / rig / RCCL paths,
# then bench.py's distributed forms.  usage: tools/dist_lines.sh <tag>
tag=$1
python -m pytest tests/test_shard_stream.py tests/test_sharded.py -m gpu -x -q > gpurun_out/${tag}_pytest_dist.log 2>&1 || { tail -n 30 gpurun_out/${tag}_pytest_dist.log; exit 1; }
tail -n 1 gpurun_out/${tag}_pytest_dist.log
python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu > gpurun_out/${tag}_rig_world1_s20_w5.json 2> gpurun_out/${tag}_dist.err || { tail gpurun_out/${tag}_dist.err; exit 1; }
python bench.py --gpus 1 --force-dist --mode sharded --steps 20 --warmup 5 --no-cpu > gpurun_out/${tag}_sharded_world1_s20_w5.json 2>> gpurun_out/${tag}_dist.err || { tail gpurun_out/${tag}_dist.err; exit 1; }
python bench.py --gpus 1 --force-dist --mode sharded --steps 100 --warmup 10 --no-cpu > gpurun_out/${tag}_sharded_world1_s100_w10.json 2>> gpurun_out/${tag}_dist.err || { tail gpurun_out/${tag}_dist.err; exit 1; }
SM_ASSOC_PAIR=0 python bench.py --gpus 1 --force-dist --mode sharded --steps 100 --warmup 10 --no-cpu > gpurun_out/${tag}_sharded_world1_s100_w10_nopair.json 2>> gpurun_out/${tag}_dist.err
for f in rig_world1_s20_w5 sharded_world1_s20_w5 sharded_world1_s100_w10 sharded_world1_s100_w10_nopair; do python - $tag $f <<PY
import json,sys
d=json.load(open("gpurun_out/%s_%s.json"%(sys.argv[1],sys.argv[2])))
print(sys.argv[2], round(d["value"]), round(d["ms_per_step"]*1e3,2), {k:v for k,v in d.items() if k in ("plain_single_gpu","vs_plain","scaling")}, str(d.get("config",{}))[:200])
PY
done
