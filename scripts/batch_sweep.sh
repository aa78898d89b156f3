# dev tool: wall time of the config-5 batch against the number of instances in flight and HIP's hardware-queue count
for q in "" 16; do for c in 1 8 16 32; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  SDPLR_BATCH_CONCURRENCY=$c timeout -k 10 120 python scripts/run_batch.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues=${q:-default} conc=$c wall_s=%.3f build_s=%.3f' % (d['wall_s'], d['problem_build_s']))"
done; done
