#!/bin/bash
# usage: prof.sh <tag>
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_trace.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_pmc1.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU -d $OUT/pmc2 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_pmc2.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_pmc3.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_pmc4.txt 2>&1
find $OUT -name "*.csv" | head -30
