#!/bin/bash
# quick GPU check: tests selected by $1, then the bench line's kernel table
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "$1" > gpurun_out/t1.log 2>&1
echo "tests rc=$?"; tail -n 4 gpurun_out/t1.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 ${@:2} > gpurun_out/b1.json 2> gpurun_out/b1.err || { tail -5 gpurun_out/b1.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/b1.json"))
print(round(d["value"]), round(d["ms_per_step"]*1e3,1), d["config"]["replay_matches_eager"], {k:v["avg_us"] for k,v in d["kernels"].items()})
PY
