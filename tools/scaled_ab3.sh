for SV in 0 1 0 1; do
ZKIR_MEASURE_SCALED=$SV timeout -k 10 300 python tools/measure_configs.py 2048 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 2048 circuits scaled=%d: %.2f ms, %.2f G checks/s, frac %.4f, scaling products %d'%(d['scaled_values'],d['gpu_ms'],d['checks_per_s']/1e9,d['mulhi_frac'],d['scaling_products']))" || exit 1
ZKIR_BENCH_SCALED_VALUES=$SV timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --linear-terms 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-term headline scaled=$SV: %.2f ms frac %.4f'%(d['ms_per_step'],d['roofline']['frac']))" || exit 1
done
