#!/bin/bash
# dev tool: kernel trace of the Lovász-θ G1–G9 lockstep batch (shared launches of the edge-path group, k_group.h):
# per-kernel launches / medians and the gaps between consecutive launches.  Writes gpurun_out/ktrace_group/summary.txt.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_group
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/probes/lovasz_group.py ${1:-3} > $OUT/log.txt 2>&1
rm -f $OUT/t_kernel_trace_keep.csv
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import sys, glob, csv, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
print("#", open(sys.argv[1] + "/log.txt").read().strip().splitlines()[-2:])
d = collections.defaultdict(list)
gaps = collections.defaultdict(list)
prev_end, prev_name = None, None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d[name].append((e - s) / 1e3)
    if prev_end is not None: gaps[name].append((s - prev_end) / 1e3)
    prev_end, prev_name = e, name
mid = [i for i, r in enumerate(rows) if "k_grp_boundary" in r["Kernel_Name"]]
if mid:
    i0 = mid[len(mid) // 2]
    t0 = int(rows[i0]["Start_Timestamp"])
    print("# timeline of 24 consecutive launches (start_us, duration_us, kernel)")
    for r in rows[i0:i0 + 24]:
        print("#  %9.2f %7.2f %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
print("kernel,launches,median_us,mean_us,total_ms,median_gap_before_us")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:24]:
    g = gaps.get(k, [0.0])
    print(f"{k},{len(v)},{statistics.median(v):.2f},{sum(v)/len(v):.2f},{sum(v)/1e3:.3f},{statistics.median(g):.2f}")
PY
find $OUT -name "*.csv" -size +2M -delete
