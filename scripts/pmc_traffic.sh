#!/bin/bash
# HBM-side traffic per launch (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes) and the kernel
# trace summary of the same bench command.  Writes gpurun_out/prof/{pmc_traffic.json,pmc_traffic.csv,
# kernel_stats.csv,kernel_medians.csv}; copy what should be judged into profiles/.
#   gpurun -- 'bash scripts/pmc_traffic.sh'
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  SDPLR_BENCH_PREWARM_S=0 rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-other-configs > $OUT/$C.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > $OUT/bench_traced.json 2> $OUT/trace.log || exit 1
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections, statistics, json, shutil
out = sys.argv[1]
SCOPE = {"k_spmm_tile": "spmm_W", "k_spmm_fast": "spmm_W", "k_lbfgs_update": "lbfgs_update", "k_lbfgs_dir": "lbfgs_dir",
         "k_fast_step2": "fast_step", "k_fast_step_ring": "fast_step", "k_lbfgs_dir_ring": "lbfgs_dir", "k_lbfgs_boundary": "lbfgs_boundary", "k_ls_solve_fast": "ls_solve_fast"}
def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0]
vals = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{C}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == C:
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    vals[C] = acc
rows, js = [], {}
for k in sorted(set(vals["FETCH_SIZE"]) | set(vals["WRITE_SIZE"])):
    fv, wv = vals["FETCH_SIZE"].get(k, [0.0]), vals["WRITE_SIZE"].get(k, [0.0])
    # real launches only: fall-through launches after loop exit move (almost) nothing
    fm = statistics.median([x for x in fv if x > 0.05 * max(fv)] or [0.0])
    wm = statistics.median([x for x in wv if x > 0.05 * max(wv)] or [0.0])
    fetch_b, write_b = 2.0 * fm * 1024.0, wm * 1024.0     # KiB; FETCH_SIZE doubled (gfx950 correction)
    rows.append((k, len(fv), fetch_b / 1e6, write_b / 1e6, (fetch_b + write_b) / 1e6))
    if k in SCOPE:
        js[SCOPE[k]] = int(fetch_b + write_b)
with open(f"{out}/pmc_traffic.csv", "w") as f:
    f.write("# HBM-side traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)\n")
    f.write("# command: SDPLR_BENCH_PREWARM_S=0 rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-other-configs\n")
    f.write("# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half of a wide coalesced read (MI355X_MICROARCH.md §HBM) and is doubled here.\n")
    f.write("# median over real launches (fall-through launches after loop exit dropped). MB = 1e6 bytes.\n")
    f.write("kernel,launches,fetch_MB_x2,write_MB,total_MB\n")
    for r in sorted(rows, key=lambda r: -r[4]):
        f.write(f"{r[0]},{r[1]},{r[2]:.2f},{r[3]:.2f},{r[4]:.2f}\n")
json.dump(js, open(f"{out}/pmc_traffic.json", "w"), indent=1)
# kernel trace
f = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[row["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
with open(f"{out}/kernel_medians.csv", "w") as g:
    g.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs ; durations in us\n")
    g.write("kernel,launches,median_us,mean_us,min_us,total_ms\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        g.write(f"{k},{len(v)},{statistics.median(v):.2f},{sum(v)/len(v):.2f},{min(v):.2f},{sum(v)/1e3:.3f}\n")
st = glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True)
if st: shutil.copy(st[0], f"{out}/kernel_stats.csv")
print(open(f"{out}/pmc_traffic.csv").read())
print(open(f"{out}/kernel_medians.csv").read()[:1800])
PY
