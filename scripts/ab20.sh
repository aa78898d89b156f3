#!/bin/bash
# A/B at the driver's protocol: --steps 20 --warmup 5
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    SDPLR_HIP_LIBRARY=$PWD/$L timeout -k 10 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$L'.split('/')[-1], round(d['value'],1))"
  done
done
