#!/bin/bash
# Kernel trace + HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of one BASELINE configuration
# that is not the headline: writes gpurun_out/prof_$1/{kernel_medians.csv,pmc_traffic.csv}; copy into profiles/.
#   gpurun -- 'bash scripts/pmc_config.sh lovasz'
W=${1:-lovasz}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$W
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/config_iters.py $W > $OUT/$C.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/config_iters.py $W > $OUT/trace.log 2>&1 || exit 1
python3 - "$OUT" "$W" <<'PY'
import sys, glob, csv, collections, statistics
out, which = sys.argv[1], sys.argv[2]
short = lambda name: name.split("(")[0].replace("void ", "")
vals = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{C}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == C:
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    vals[C] = acc
rows = []
for k in sorted(set(vals["FETCH_SIZE"]) | set(vals["WRITE_SIZE"])):
    fv, wv = vals["FETCH_SIZE"].get(k, [0.0]), vals["WRITE_SIZE"].get(k, [0.0])
    fm = statistics.median([x for x in fv if x > 0.05 * max(fv)] or [0.0])
    wm = statistics.median([x for x in wv if x > 0.05 * max(wv)] or [0.0])
    fetch_b, write_b = 2.0 * fm * 1024.0, wm * 1024.0     # KiB; FETCH_SIZE doubled (gfx950 correction)
    rows.append((k, len(fv), fetch_b / 1e6, write_b / 1e6, (fetch_b + write_b) / 1e6))
with open(f"{out}/pmc_traffic.csv", "w") as f:
    f.write(f"# HBM-side traffic per launch, configuration `{which}` (scripts/config_iters.py): rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes\n")
    f.write("# FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md §HBM). median over real launches. MB = 1e6 bytes.\n")
    f.write("kernel,launches,fetch_MB_x2,write_MB,total_MB\n")
    for r in sorted(rows, key=lambda r: -r[4]):
        f.write(f"{r[0]},{r[1]},{r[2]:.2f},{r[3]:.2f},{r[4]:.2f}\n")
f = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
with open(f"{out}/kernel_medians.csv", "w") as g:
    g.write(f"# rocprofv3 --kernel-trace --stats -- python3 scripts/config_iters.py {which} ; durations in us\n")
    g.write("kernel,launches,median_us,mean_us,min_us,total_ms\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        g.write(f"{k},{len(v)},{statistics.median(v):.2f},{sum(v)/len(v):.2f},{min(v):.2f},{sum(v)/1e3:.3f}\n")
print(open(f"{out}/pmc_traffic.csv").read()[:2500])
print(open(f"{out}/kernel_medians.csv").read()[:2500])
PY
