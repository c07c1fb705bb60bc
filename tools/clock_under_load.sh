#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/clk
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/a -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/a.txt 2>&1
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/g -o run -- python3 bench.py --workload gadget --log2-vectors 19 --gadget-rounds 2000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/g.txt 2>&1
echo done
