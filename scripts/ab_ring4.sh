# A/B of the ring form of the history (SDPLR_HIP_NO_RING=1: stored form) on BASELINE configs 3 and 4 (scripts/configs_bench.py)
for i in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export SDPLR_HIP_NO_RING=1; else unset SDPLR_HIP_NO_RING; fi
    timeout -k 10 300 python scripts/configs_bench.py 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print('noring=$v', 'lovasz', round(d['config3_lovasz_theta']['inner_iterations_per_s'],1), 'minbis', round(d['config4_minimum_bisection']['inner_iterations_per_s'],1))"
  done
done
