#!/bin/bash
for d in 0 1 3 4 8; do
FC3D_DBG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --launch eager > gpurun_out/bd.json 2> gpurun_out/bd.err || { tail -5 gpurun_out/bd.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bd.json"))
print("dbg=$d", d["kernels"]["bilinear_bwd_img_gather"]["avg_us"], d["kernels"]["warp_sample"]["avg_us"])
PY
done
