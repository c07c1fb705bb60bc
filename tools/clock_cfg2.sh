#!/bin/bash
# GRBM_GUI_ACTIVE (GPU clock cycles) over the headline launch: usage clock_cfg2.sh <tag> [bench.py flags]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/clk_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/a.txt 2>&1
echo done
