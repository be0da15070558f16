#!/bin/bash
# Bench lines (and rocprofv3 kernel stats for C4 / C5) of the configurations other than the headline one:
#   profiles/collect_configs.sh <tag>
# Results land in gpurun_out/<tag>_*; copy what is to be kept into profiles/.
set -u
tag=${1:-r03b}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
mkdir -p gpurun_out
run() {   # name, bench arguments...
  local name=$1; shift
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err || { echo "$name failed"; tail -3 gpurun_out/${tag}_${name}.err; exit 1; }
  echo "$name ok"
}
run c1 --workload C1 --cpu-seconds 6
run c5 --workload C5 --no-cpu-baseline
run c4_k23 --workload C4 --no-cpu-baseline
run c4_k33 --workload C4 --psf-k 33 --no-cpu-baseline
run c2_k3 --psf-k 3 --no-cpu-baseline
run c2_k7 --psf-k 7 --no-cpu-baseline
run c2_k9 --psf-k 9 --no-cpu-baseline
run c2_k11 --psf-k 11 --no-cpu-baseline
for w in C4 C5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${w}_stats -- \
      python3 bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${tag}_${w}_stats.log 2>&1 || { echo "$w stats failed"; exit 1; }
  echo "$w stats ok"
done
