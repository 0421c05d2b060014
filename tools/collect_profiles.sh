#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/prof_<tag>_{s20,s100,hd} into profiles/ under the round's name:
# kernel_stats csv, prof_summary.py's table, the traffic json bench.py reads, the bench line printed under the profiler.
# usage: tools/collect_profiles.sh <tag, e.g. r03a> <round dir, e.g. r03>
set -e
tag=$1; rd=$2
mkdir -p profiles/$rd
for t in s20:20:5:kitti_s20_w5 s100:100:10:kitti_s100_w10 hd:40:5:hd20m_s40_w5; do
    IFS=: read n st w name <<< "$t"
    d=gpurun_out/prof_${tag}_$n
    [ -d $d ] || continue
    python tools/prof_summary.py $d --last $st --steps $st --warmup $w --out profiles/traffic_$name.json > profiles/${rd}_summary_$n.txt
    cp $d/trace/*/*_kernel_stats.csv profiles/$rd/kernel_stats_$n.csv
    cp $d.bench_under_profiler.json profiles/${rd}_bench_under_profiler_$n.json
done
ls profiles/$rd
