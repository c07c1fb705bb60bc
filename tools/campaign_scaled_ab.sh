#!/bin/bash
# Configuration 4 on one GPU: campaign circuits lowered with and without scaled values (same box, alternating).
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["also"][-1]
print(sys.argv[1], "circuits/s %.0f  G checks/s %.2f  seconds %.3f"%(e["circuits_per_s"], e["value"]/1e9, e["seconds"]), {k:round(v,3) for k,v in e["phases_rank0"].items()})'
for rep in 1 2 3; do
timeout -k 10 200 $B 2>/dev/null | python -c "$P" "scaled_values=0" || exit 1
ZKIR_CAMPAIGN_LOWER="scaled_values=1" timeout -k 10 200 $B 2>/dev/null | python -c "$P" "scaled_values=1" || exit 1
done
