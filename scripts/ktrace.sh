#!/bin/bash
# dev tool: rocprofv3 kernel trace of a short bench run; prints per-kernel median / mean durations
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${1:-200} --warmup 20 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -c 300 $OUT/bench.log | head -c 0
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[row["Kernel_Name"].split("(")[0][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{k:62s} n={len(v):6d} median={statistics.median(v):8.2f} mean={sum(v)/len(v):8.2f} us  {100*sum(v)/tot:5.1f}%")
PY
