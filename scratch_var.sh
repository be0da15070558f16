#!/bin/bash
# bench every variant library under scratch/libs (or those named), print kernel table
for lib in ${@:-scratch/libs/*.so}; do
FC3D_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 ${BENCH_ARGS} > gpurun_out/bv.json 2> gpurun_out/bv.err || { tail -5 gpurun_out/bv.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bv.json"))
print("$lib", round(d["value"]), round(d["ms_per_step"]*1e3,1), {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
done
