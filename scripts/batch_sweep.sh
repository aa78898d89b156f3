# dev tool: wall time of the config-5 batch against the number of instances in flight
for c in 4 8 12 16 24; do
  SDPLR_BATCH_CONCURRENCY=$c timeout -k 10 120 python scripts/run_batch.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conc=$c wall_s=%.3f build_s=%.3f' % (d['wall_s'], d['problem_build_s']))"
done
