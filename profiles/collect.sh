#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers, on the GPU box:
#   profiles/collect.sh <tag>
# 1. kernel trace + stats of the default bench command (per-kernel average durations),
# 2. one counter pass for FETCH_SIZE, one for WRITE_SIZE, on a short run of the
#    same step (counter passes serialise kernels, so they are never used for timing).
# Results land in gpurun_out/<tag>_*; profiles/make_traffic.py turns them into the
# committed summaries.
set -u
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- \
    python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1 || { echo "stats pass failed"; exit 1; }
echo "stats ok"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- \
    python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${tag}_pmc_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
echo "fetch ok"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- \
    python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${tag}_pmc_write.log 2>&1 || { echo "write pass failed"; exit 1; }
echo "write ok"
