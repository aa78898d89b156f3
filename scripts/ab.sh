#!/bin/bash
# dev tool: A/B two builds of the HIP library on the same box, alternating runs
#   gpurun -- 'bash scripts/ab.sh path/A.so path/B.so [rounds]'
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    SDPLR_HIP_LIBRARY=$PWD/$L timeout -k 10 150 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$L'.split('/')[-1], round(d['value'],1), {k:v['us_per_step'] for k,v in d['kernels_eager_profile'].items() if v['us_per_step']>3})"
  done
done
