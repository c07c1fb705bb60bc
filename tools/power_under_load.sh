#!/bin/bash
# usage: power_under_load.sh: socket power and shader clock sampled by rocm-smi while bench.py repeats the headline
# launch, and while it repeats the gadget launch (-> gpurun_out/power.txt)
OUT=$GRAFT_REPO_ROOT/gpurun_out/power.txt
: > $OUT
sample() {  # $1: label; samples until the file gpurun_out/.stop appears
  while [ ! -e $GRAFT_REPO_ROOT/gpurun_out/.stop ]; do
    echo "$1 $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' ';')" >> $OUT
    sleep 0.5
  done
}
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/.stop
echo "idle $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' ';')" >> $OUT
sample headline & SP=$!
timeout -k 10 150 python3 bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/power_headline.json 2>/dev/null
touch gpurun_out/.stop; wait $SP; rm -f gpurun_out/.stop
sample gadget & SP=$!
timeout -k 10 150 python3 bench.py --workload gadget --log2-vectors 19 --gadget-rounds 2000 --steps 30 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/power_gadget.json 2>/dev/null
touch gpurun_out/.stop; wait $SP; rm -f gpurun_out/.stop
sort $OUT | uniq -c | sort -rn | head -40
