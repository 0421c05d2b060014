#!/bin/bash
# Full GPU validation + profile refresh of one round (run on the MI355X box through gpurun): tests, smoke, the three bench
# lines, then tools/profile_round.sh for the driver's configuration, the 100-frame run and configs[2].  usage: tools/round_final.sh <tag>
tag=$1
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1 || { tail -n 30 gpurun_out/${tag}_pytest_gpu.log; exit 1; }
tail -n 1 gpurun_out/${tag}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 || { tail gpurun_out/${tag}_smoke.log; exit 1; }
tail -n 1 gpurun_out/${tag}_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_kitti_s20_w5.json 2> gpurun_out/${tag}_bench.err || exit 1
python bench.py --steps 100 --warmup 10 > gpurun_out/${tag}_bench_kitti_s100_w10.json 2>> gpurun_out/${tag}_bench.err || exit 1
python bench.py --workload hd20m --steps 40 --warmup 5 > gpurun_out/${tag}_bench_hd20m.json 2>> gpurun_out/${tag}_bench.err || exit 1
echo "bench lines done"
bash tools/profile_round.sh prof_${tag}_s20 20 5 && bash tools/profile_round.sh prof_${tag}_s100 100 10 && bash tools/profile_round.sh prof_${tag}_hd 40 5 --workload hd20m
