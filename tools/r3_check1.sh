#!/bin/bash
# round 3, first GPU check: the changed paths, then the default bench line, then the full-size configs
tag=r3a
timeout -k 10 500 python -m pytest tests/test_rig.py tests/test_shard_stream.py tests/test_deferred_compaction.py tests/test_kat.py -m gpu -x -q > gpurun_out/${tag}_pytest1.log 2>&1; rc=$?
tail -n 5 gpurun_out/${tag}_pytest1.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; rc=$?
echo "bench rc=$rc"; tail -n 3 gpurun_out/${tag}_bench_default.err
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 > gpurun_out/${tag}_bench_ranks1.json 2> gpurun_out/${tag}_bench_ranks1.err; rc=$?
echo "bench ranks rc=$rc"; tail -n 3 gpurun_out/${tag}_bench_ranks1.err
[ $rc -ne 0 ] && exit 1
timeout -k 10 1100 python -m pytest tests/test_configs_full_size.py -m gpu -x -q > gpurun_out/${tag}_pytest_full.log 2>&1; rc=$?
tail -n 5 gpurun_out/${tag}_pytest_full.log
exit $rc
