#!/bin/bash
# Collects the judged profile artefacts of one round into gpurun_out/profiles_<tag>/ (copy to profiles/ afterwards):
#   <tag>_bench_kernel_stats.csv   per-kernel statistics of `python bench.py` from the rocprofv3 kernel trace, counting
#                                   only each kernel's DOMINANT launch shape (set-up launches such as the one-image LCN
#                                   of the pattern have another grid) and of those the last 20 = the timed steps (the
#                                   settle loop before them runs on rising clocks; its average is kept in a column of its own)
#   <tag>_pmc_hbm_traffic.json     FETCH_SIZE / WRITE_SIZE (KiB) per launch of the same launches, separate --pmc passes
#   <tag>_bench_under_rocprof.json the bench line of the traced run
# bench.py runs with --headline-only: the parity probe's 1-frame launches and the also_measured legs (plain volume
# kernel, volume-free path) would be averaged into the same kernels' rows.
tag=${1:-round2}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --headline-only --workload config2"
# 1. kernel trace of the exact bench command (no PMC in this pass)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 20 --warmup 3 > $out/bench_under_rocprof.log 2>&1 || exit 1
grep '^{"metric"' $out/bench_under_rocprof.log > $out/${tag}_bench_under_rocprof.json
# 2. HBM traffic counters, separate passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- $B --steps 4 --warmup 1 > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $out/pmc_write -- $B --steps 4 --warmup 1 > $out/pmc_write.log 2>&1 || exit 1
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]

def short(name):
    return name.split("(")[0]

# ---- kernel statistics per (kernel, grid), dominant grid only
trace = glob.glob(out + "/trace/*/*kernel_trace.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(trace)):
    grid = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    per[short(r["Kernel_Name"])][grid].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
STEPS = 20          # the timed region of the traced command: the LAST `--steps` launches of every kernel's dominant shape
for k, grids in per.items():
    grid, d_all = max(grids.items(), key=lambda kv: sum(kv[1]))
    d = d_all[-STEPS:]          # (trace rows are in dispatch order; the settle loop and warm-up before them run on rising clocks)
    n = len(d)
    mean = sum(d) / n
    sd = (sum((x - mean) ** 2 for x in d) / n) ** 0.5
    rows.append((sum(d), k, "x".join(grid), n, mean, min(d), max(d), sd, sum(len(v) for v in grids.values()) - n))
    rows[-1] = rows[-1] + (sum(d_all) / len(d_all), len(d_all))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
with open("%s/%s_bench_kernel_stats.csv" % (out, tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Grid", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "LaunchesLeftOut",
                "AverageNsAllLaunchesOfTheRun", "AllLaunchesOfTheRun"])
    for tot, k, grid, n, mean, lo, hi, sd, other, mean_all, n_all in rows:
        w.writerow([k, grid, n, tot, "%.1f" % mean, "%.2f" % (100.0 * tot / total), lo, hi, "%.1f" % sd, other, "%.1f" % mean_all, n_all])
        print("%-60s grid %-16s timed %3d avg %9.1f us  (all %d launches of the run: %.1f us)" % (k[:60], grid, n, mean / 1e3, n_all, mean_all / 1e3))

# ---- HBM traffic of the same launch shapes
dominant = {k: g for _, k, g, *_ in rows}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    tr = glob.glob(f.rsplit("/", 1)[0] + "/*kernel_trace.csv")
    grids = {}
    if tr:
        for r in csv.DictReader(open(tr[0])):
            grids[r["Dispatch_Id"]] = "x".join((r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if "ctd::" not in k or grids.get(row["Dispatch_Id"], dominant.get(k)) != dominant.get(k):
            continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
summ = {}
for k, cs in agg.items():
    summ[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    summ[k]["launches_sampled"] = max(len(v) for v in cs.values())
    summ[k]["grid"] = dominant.get(k)
    if "FETCH_SIZE" in summ[k] and "WRITE_SIZE" in summ[k]:
        # MI355X_MICROARCH.md: FETCH_SIZE under-reports by 2x on gfx950, WRITE_SIZE is exact; both in KiB
        summ[k]["hbm_bytes_per_launch"] = (2.0 * summ[k]["FETCH_SIZE"] + summ[k]["WRITE_SIZE"]) * 1024.0
json.dump(summ, open("%s/%s_pmc_hbm_traffic.json" % (out, tag), "w"), indent=1)
print(json.dumps({k: v for k, v in summ.items() if "t256" in k or "alld" in k}, indent=1))
PY
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
