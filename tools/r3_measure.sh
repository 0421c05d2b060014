#!/bin/bash
# end-of-round measurement: rocprofv3 passes at the three profiled configurations, then (the traffic files it leaves under
# gpurun_out are NOT yet in profiles/: run tools/collect_profiles.sh and this script's second half again if the period changed)
# the bench lines and the launch traces.  usage: tools/r3_measure.sh <tag> [lines-only]
tag=${1:-r03}
if [ "$2" != "lines-only" ]; then
  timeout -k 10 1000 bash tools/r3_prof.sh ${tag} || exit 1
fi
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${tag}_kitti_s20_w5.json 2> gpurun_out/${tag}_lines.err || exit 1
python bench.py > gpurun_out/bench_${tag}_default.json 2>> gpurun_out/${tag}_lines.err || exit 1
python bench.py --steps 100 --warmup 10 --only-headline > gpurun_out/bench_${tag}_kitti_s100_w10.json 2>> gpurun_out/${tag}_lines.err || exit 1
python bench.py --workload hd20m --steps 40 --warmup 5 --only-headline > gpurun_out/bench_${tag}_hd20m.json 2>> gpurun_out/${tag}_lines.err || exit 1
python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > gpurun_out/bench_${tag}_ranks_world1.json 2>> gpurun_out/${tag}_lines.err || exit 1
python tools/pass_trace.py 23 > gpurun_out/${tag}_pass_trace_kitti_frame22.txt 2>&1 && python tools/pass_trace.py 110 > gpurun_out/${tag}_pass_trace_kitti_frame109.txt 2>&1
echo measured
