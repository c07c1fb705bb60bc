# usage: scaled_ab.sh: the headline launch and the earlier rounds' circuit with every value plain and with scaled values
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]
print(sys.argv[1], "ms %.2f  frac %.4f  G checks/s %.2f  ops %d gslots %d scaling products %d"%(d["ms_per_step"], d["roofline"]["frac"], d["value"]/1e9, c["n_ops"], c["device_slots"], c["n_scaling_products"]))'
for rep in 1 2; do
for SV in 0 1; do
ZKIR_BENCH_SCALED_VALUES=$SV timeout -k 10 200 $B 2>/dev/null | python -c "$P" "headline scaled_values=$SV" || exit 1
ZKIR_BENCH_SCALED_VALUES=$SV timeout -k 10 200 $B --constant-factors 2>/dev/null | python -c "$P" "constant-factors scaled_values=$SV" || exit 1
done
done
