# bench.py inside a torch.distributed process (torch's bundled HIP runtime serves the library): graph batches, eager launches, longer graphs
export SDPLR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
port=29520
for v in graph eager g32 graph eager; do
  unset SDPLR_HIP_NO_GRAPH SDPLR_HIP_GRAPH_ITERS
  if [ $v = eager ]; then export SDPLR_HIP_NO_GRAPH=1; fi
  if [ $v = g32 ]; then export SDPLR_HIP_GRAPH_ITERS=32; fi
  port=$((port+1)); export MASTER_PORT=$port
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
  python -c "
import json;d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1]);print('$v', round(d['value'],1))" || tail -3 /tmp/o.err
done
