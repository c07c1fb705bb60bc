#!/bin/bash
# Priority turns in the 8 x 32-bit stream: gadget strand tapes forced onto that stream, with and without turns.
set -o pipefail
out=gpurun_out/r3o_turns32.txt
: > $out
run() {  # label, env...
  echo "== $1" >> $out; shift
  env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --workload gadget --gadget-rounds 4000 --log2-vectors $LOG --steps 5 --warmup 2 >> $out 2>&1 || exit 1
}
for LOG in 15 16 17; do
  for rep in 1 2; do
    run "lean32 no turns 2^$LOG" ZKIR_INTERP=lean32 ZKIR_TURN_SHIFT=99 || exit 1
    run "lean32 turns 2^$LOG" ZKIR_INTERP=lean32 || exit 1
  done
  run "lean29 turns 2^$LOG" ZKIR_INTERP=lean29 || exit 1
done
grep -E "^==|ms_per_step" $out | sed -E 's/.*"ms_per_step": ([0-9.]+).*/   \1 ms/' 
