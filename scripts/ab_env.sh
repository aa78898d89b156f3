#!/bin/bash
# A/B of one environment knob at the driver's protocol: bash scripts/ab_env.sh VAR "v1 v2 v3" [repeats] [steps]
VAR=$1; VALS=$2; R=${3:-2}; K=${4:-20}
for i in $(seq $R); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 150 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_eager_profile'];print('$VAR=$v', round(d['value'],1), {n:k[n]['us_per_step'] for n in ('fast_step','spmm_W','lbfgs_dir','lbfgs_boundary') if n in k})"
  done
done
