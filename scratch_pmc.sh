#!/bin/bash
# SQ counter pass over a short eager run of the step: per-kernel wave cycles, parked cycles, issue counts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_sq
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --launch eager ${BENCH_ARGS} > gpurun_out/pmc_sq.log 2>&1 || { echo "pmc pass failed"; tail -5 gpurun_out/pmc_sq.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void fc3d::", "").replace("fc3d::", "")[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1
print("%-46s %5s %10s %7s %7s %7s %9s %9s %8s" % ("kernel", "n", "wavecyc", "wait%", "winst%", "act%", "valu/wv", "salu/wv", "lds/wv"))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:14]:
    wc = c["SQ_WAVE_CYCLES"] or 1
    print("%-46s %5d %10.3g %7.1f %7.1f %7.1f %9.3g %9.3g %8.3g" % (k, n[k], wc / n[k], 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
          100 * c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_INSTS_VALU"] / n[k], c["SQ_INSTS_SALU"] / n[k], c["SQ_INSTS_LDS"] / n[k]))
PY
