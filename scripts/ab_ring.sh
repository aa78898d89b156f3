for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export SDPLR_HIP_NO_RING=1; else unset SDPLR_HIP_NO_RING; fi
    timeout -k 10 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('noring=$v steps20', round(d['value'],1), d.get('roofline',{}).get('achieved'))"
  done
done
for v in 0 1; do
  if [ $v = 1 ]; then export SDPLR_HIP_NO_RING=1; else unset SDPLR_HIP_NO_RING; fi
  timeout -k 10 150 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('noring=$v steps300', round(d['value'],1), json.dumps(d.get('roofline')), d.get('parity'))"
done
