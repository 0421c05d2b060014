#!/bin/bash
# The multi-GPU modes rehearsed with one rank on the GPU box (torch.distributed + RCCL up at world = 1): tests of the sharded /
# rig / RCCL paths, then bench.py's N-rank form (rig value + consolidation + sharded_leg in one line).  usage: tools/dist_lines.sh <tag>
tag=$1
python -m pytest tests/test_shard_stream.py tests/test_sharded.py tests/test_rig.py tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/${tag}_pytest_dist.log 2>&1 || { tail -n 30 gpurun_out/${tag}_pytest_dist.log; exit 1; }
tail -n 1 gpurun_out/${tag}_pytest_dist.log
python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > gpurun_out/${tag}_ranks_world1_s20_w5.json 2> gpurun_out/${tag}_dist.err || { tail gpurun_out/${tag}_dist.err; exit 1; }
python bench.py --gpus 1 --force-dist --steps 100 --warmup 10 > gpurun_out/${tag}_ranks_world1_s100_w10.json 2>> gpurun_out/${tag}_dist.err || { tail gpurun_out/${tag}_dist.err; exit 1; }
for f in ranks_world1_s20_w5 ranks_world1_s100_w10; do python - $tag $f <<PY
import json,sys
d=json.load(open("gpurun_out/%s_%s.json"%(sys.argv[1],sys.argv[2])))
sl=d.get("sharded_leg") or {}
print(sys.argv[2], "rig", round(d["value"]), round(d["ms_per_step"]*1e3,2), "us | consolidation ms", round(d["config"]["multi_gpu"]["consolidation_ms"],2),
      "| sharded", sl.get("value") and round(sl["value"]), "x plain", sl.get("sharded_over_plain") and round(sl["sharded_over_plain"],3), "| rccl", d.get("rccl",{}).get("nranks"))
PY
done
