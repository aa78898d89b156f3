#!/bin/bash
# dev tool: rocprofv3 kernel trace of scripts/family_rate.py <maxcut|minbis|lovasz>: per-kernel medians
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_fam
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/family_rate.py ${1:-minbis} > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:9]:
    print(f"{k:52s} n={len(v):5d} median={statistics.median(v):7.2f} us")
PY
