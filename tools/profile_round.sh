#!/bin/bash
# Profile one bench.py command with rocprofv3 (run on the MI355X box through gpurun): kernel trace + the three PMC passes,
# each in its own run (profiles/README.md).  usage: tools/profile_round.sh <outdir under gpurun_out> <steps> <warmup> [extra bench args]
set -e
export TMPDIR=/tmp
out=gpurun_out/$1; steps=$2; warm=$3; shift 3
args="--steps $steps --warmup $warm --workers 1 --no-cpu-baseline --only-headline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out.bench_under_profiler.json 2> $out.trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $args > /dev/null 2> $out.fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $args > /dev/null 2> $out.write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES \
          --output-format csv -d $out/pmc_sq -- python3 bench.py $args > /dev/null 2> $out.sq.err
echo "profiled: $out"
