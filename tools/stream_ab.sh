# same box: the library built with the previous instruction stream (scratch/libzkir_oldstream.so) against the current
# one, same tapes (OUTPUT before CHECK, so that the current stream's operand reuse never triggers), then the current
# one with the OUTPUT op behind the CHECK
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1], "ms %.2f  frac %.4f  ops %d gslots %d"%(d["ms_per_step"], d["roofline"]["frac"], c["n_ops"], c["device_slots"]))'
for rep in 1 2 3; do
for W in "" "--constant-factors"; do
ZKIR_LIB=$PWD/scratch/libzkir_oldstream.so ZKIR_LOWER_OUTPUT_FIRST=1 timeout -k 10 200 $B $W 2>/dev/null | python -c "$P" "old stream, output first $W" || exit 1
ZKIR_LOWER_OUTPUT_FIRST=1 timeout -k 10 200 $B $W 2>/dev/null | python -c "$P" "new stream, output first $W" || exit 1
timeout -k 10 200 $B $W 2>/dev/null | python -c "$P" "new stream, output last  $W" || exit 1
done
done
