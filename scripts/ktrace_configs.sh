#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_lv
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/configs_bench.py > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# keep only the first 60% of the trace (config 3 runs first) by time
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
d = collections.defaultdict(list)
seen_tile = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_spmm_tile" in name and seen_tile is None: seen_tile = int(r["Start_Timestamp"])
for r in rows:
    if seen_tile is not None and int(r["Start_Timestamp"]) >= seen_tile: continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1]))[:22]:
    print(f"{k:40s} n={len(v):6d} median={statistics.median(v):8.2f} us")
PY
