for rep in 1 2; do for v in old new; do
  lib=surfelmapping_amd/libsurfelmapping_hip.so; [ $v = old ] && lib=surfelmapping_amd/libsurfelmapping_hip_old.so
  SM_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-fuse-leg --no-steady-leg --no-hd-leg --no-cpu-baseline > gpurun_out/abc_${v}_$rep.json 2>> gpurun_out/abc.err || exit 1
  python - gpurun_out/abc_${v}_$rep.json $v <<PY
import json,sys
d=json.load(open(sys.argv[1])); r=d["reference_path_leg"]
print(sys.argv[2], {k:(round(v["value"]), round(v["ms_per_step"]*1e3,1)) for k,v in r.items() if isinstance(v,dict)}, "headline", round(d["value"]))
PY
done; done
SM_HIP_LIB=$PWD/surfelmapping_amd/libsurfelmapping_hip_old.so python tools/chain_trace.py > gpurun_out/abc_trace_old.txt 2>&1; tail -n 4 gpurun_out/abc_trace_old.txt
python tools/chain_trace.py > gpurun_out/abc_trace_new.txt 2>&1; tail -n 4 gpurun_out/abc_trace_new.txt
