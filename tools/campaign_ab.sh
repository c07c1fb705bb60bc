set -e
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["also"][-1]
print(sys.argv[1], "circuits/s %.0f  G checks/s %.2f  seconds %.3f"%(e["circuits_per_s"], e["value"]/1e9, e["seconds"]), {k:round(v,3) for k,v in e["phases_rank0"].items()})'
for rep in 1 2; do
for G in 512 1024 2048; do
ZKIR_CAMPAIGN_GROUP=$G timeout -k 10 200 $B 2>/dev/null | python -c "$P" "group=$G" 
done
ZKIR_CAMPAIGN_LOWER="order_ops=1,window_cache=1" ZKIR_CAMPAIGN_GROUP=512 timeout -k 10 200 $B 2>/dev/null | python -c "$P" "group=512 full-placement"
ZKIR_CAMPAIGN_GROUP=512 timeout -k 10 200 $B --campaign-circuits 16384 2>/dev/null | python -c "$P" "group=512 16384 circuits"
done
