#!/bin/bash
# usage: profiles/collect_sq_c4.sh tag [workload]  -- two SQ counter passes (8 counters each) over bench.py --workload C4 (or the one named); summary in gpurun_out/<tag>_sq.txt
tag=$1
wl=${2:-C4}
root=$PWD

cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
p1="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
p2="SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY"
i=1
for p in "$p1" "$p2"; do
  rm -rf gpurun_out/${tag}_sq$i
  timeout -s KILL 150 rocprofv3 --pmc $p --output-format csv -d gpurun_out/${tag}_sq$i -- python3 bench.py --workload $wl --launch eager --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${tag}_sq$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${tag}_sq$i.log; exit 1; }
  echo "pass $i ok"
  i=$((i+1))
done
python3 profiles/sq_parse.py gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 > gpurun_out/${tag}_sq.txt
cat gpurun_out/${tag}_sq.txt
find gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 -name "*.csv" ! -name "*counter_collection*" -delete
