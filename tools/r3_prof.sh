#!/bin/bash
# Round-3 profile refresh (run on the MI355X box through gpurun): the three profiled configurations.  usage: tools/r3_prof.sh <tag>
tag=$1
bash tools/profile_round.sh prof_${tag}_s20 20 5 && bash tools/profile_round.sh prof_${tag}_s100 100 10 && bash tools/profile_round.sh prof_${tag}_hd 40 5 --workload hd20m
