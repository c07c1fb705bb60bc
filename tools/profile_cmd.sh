#!/bin/bash
# usage: profile_cmd.sh <tag> <program and arguments...>: kernel trace + counter passes of one command
# (-> gpurun_out/prof_<tag>; summarise with rocprof_summary.py).  One counter pass per run, within the per-block limits.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- "$@" > $OUT/out_trace.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o run -- "$@" > $OUT/out_pmc1.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o run -- "$@" > $OUT/out_pmc2.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o run -- "$@" > $OUT/out_pmc3.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o run -- "$@" > $OUT/out_pmc4.txt 2>&1 || exit 1
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4  # the databases are tens of megabytes: only the summary travels back
cat $OUT/summary.txt
