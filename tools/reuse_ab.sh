# the headline launch and the earlier rounds' circuit with the OUTPUT op of a constraint behind its CHECK (the CHECK
# then reuses the operands the MUL unpacked) and in the earlier order
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1], "ms %.2f  frac %.4f  G checks/s %.2f  ops %d gslots %d scaling products %d"%(d["ms_per_step"], d["roofline"]["frac"], d["value"]/1e9, c["n_ops"], c["device_slots"], c["n_scaling_products"]))'
for rep in 1 2; do
timeout -k 10 200 $B 2>/dev/null | python -c "$P" "headline output-last" || exit 1
ZKIR_LOWER_OUTPUT_FIRST=1 timeout -k 10 200 $B 2>/dev/null | python -c "$P" "headline output-first" || exit 1
timeout -k 10 200 $B --constant-factors 2>/dev/null | python -c "$P" "constant-factors output-last" || exit 1
ZKIR_LOWER_OUTPUT_FIRST=1 timeout -k 10 200 $B --constant-factors 2>/dev/null | python -c "$P" "constant-factors output-first" || exit 1
done
