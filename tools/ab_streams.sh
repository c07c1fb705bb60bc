#!/bin/bash
# usage: ab_streams.sh <tag>: the headline launch through both generated streams (8 x 32-bit, 9 x 29-bit), with the
# VALU / LDS / clock counters of each (-> gpurun_out/ab_<tag>, summarise with rocprof_summary.py)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for K in lean32 lean29; do
  export ZKIR_INTERP=$K
  timeout -k 10 120 $B > $OUT/line_$K.json 2> $OUT/line_$K.err
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1_$K -o run -- $B > $OUT/pmc1_$K.txt 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc2_$K -o run -- $B > $OUT/pmc2_$K.txt 2>&1
done
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
grep -h ms_per_step $OUT/line_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['roofline']['kernel'], d['ms_per_step'], d['roofline']['frac'])" >> $OUT/summary.txt
cat $OUT/summary.txt
