#!/bin/bash
# dev tool: kernel trace of one small solve (Gset G1): per-kernel medians and the idle gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_small
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/small_instance_profile.py > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev_end = None; prev_name = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d[name].append((e - s) / 1e3)
    if prev_end is not None: gaps[name].append((s - prev_end) / 1e3)
    prev_end, prev_name = e, name
tot = sum(sum(v) for v in d.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print(f"kernels {len(rows)}  busy {tot/1e3:.2f} ms  span {span:.2f} ms")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:22]:
    g = gaps.get(k, [0])
    print(f"{k:46s} n={len(v):5d} median={statistics.median(v):7.2f} us  gap-before median={statistics.median(g):7.2f} us  total={sum(v)/1e3:7.3f} ms")
PY
