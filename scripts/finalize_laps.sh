SDPLR_HIP_TIMING=1 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python scripts/concurrency_probe.py 2> gpurun_out/r3_laps.err > /dev/null
python - <<'PY'
import re, collections
d = collections.defaultdict(list)
for line in open('gpurun_out/r3_laps.err'):
    m = re.match(r'\[sdplr_hip finalize\] (.*?)\s+([0-9.]+) ms', line)
    if m: d[m.group(1).strip()].append(float(m.group(2)))
for k, v in d.items():
    v1 = v[:5]; v2 = v[-64:]
    print(f"{k:32s} first(1 thread) mean {sum(v1)/len(v1):7.3f} ms   last 64 (16 threads) mean {sum(v2)/len(v2):7.3f} ms  max {max(v2):7.3f}")
PY
