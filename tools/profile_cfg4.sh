#!/bin/bash
# rocprofv3 kernel statistics of `python bench.py --workload config4` (-> gpurun_out/profiles_<tag>/, copy to profiles/)
tag=${1:-round3}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace4 -- python bench.py --workload config4 --no-cpu-baseline --steps 20 --warmup 3 > $out/cfg4_under_rocprof.log 2>&1 || exit 1
grep '^{"metric"' $out/cfg4_under_rocprof.log > $out/${tag}_config4_bench_under_rocprof.json
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
trace = glob.glob(out + "/trace4/*/*kernel_trace.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(trace)):
    grid = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    per[r["Kernel_Name"].split("(")[0]][grid].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, grids in per.items():
    grid, d = max(grids.items(), key=lambda kv: sum(kv[1]))
    n = len(d); mean = sum(d) / n
    rows.append((sum(d), k, "x".join(grid), n, mean, min(d), max(d), sum(len(v) for v in grids.values()) - n))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
with open("%s/%s_config4_kernel_stats.csv" % (out, tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Grid", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsWithOtherGrids"])
    for tot, k, grid, n, mean, lo, hi, other in rows:
        w.writerow([k, grid, n, tot, "%.1f" % mean, "%.2f" % (100.0 * tot / total), lo, hi, other])
        print("%-60s grid %-18s calls %3d avg %9.1f us" % (k[:60], grid, n, mean / 1e3))
PY
# HBM traffic of the same kernels: separate --pmc passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2), no tracing domains
B4="python bench.py --workload config4 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc4_fetch -- $B4 --steps 4 --warmup 1 > $out/pmc4_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc4_write -- $B4 --steps 4 --warmup 1 > $out/pmc4_write.log 2>&1 || exit 1
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc4_*/*/*counter_collection.csv"):
    tr = glob.glob(f.rsplit("/", 1)[0] + "/*kernel_trace.csv")
    grids = {}
    if tr:
        for r in csv.DictReader(open(tr[0])):
            grids[r["Dispatch_Id"]] = "x".join((r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "ctd::" not in k:
            continue
        agg[(k, grids.get(row["Dispatch_Id"], "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
# per kernel: the launch shape with the most LAUNCHES (the timed one-frame step and its settle loop -- not the one-image
# set-up launches, and not the two-frame call of the `ncc_two_frames_per_call` leg, which moves twice the bytes: until round 5
# the shape with the most bytes was taken, i.e. the two-frame call's)
best = {}
for (k, g), cs in agg.items():
    tot = max(len(v) for v in cs.values())
    if k not in best or tot > best[k][0]:
        best[k] = (tot, g, cs)
summ = {}
for k, (tot, g, cs) in best.items():
    summ[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    summ[k]["grid"] = g
    summ[k]["launches_sampled"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in summ[k] and "WRITE_SIZE" in summ[k]:
        summ[k]["hbm_bytes_per_launch"] = (2.0 * summ[k]["FETCH_SIZE"] + summ[k]["WRITE_SIZE"]) * 1024.0   # KiB; FETCH_SIZE x 2 on gfx950
json.dump(summ, open("%s/%s_config4_pmc_hbm_traffic.json" % (out, tag), "w"), indent=1)
print(json.dumps({k: v for k, v in summ.items() if "alld" in k or "census" in k}, indent=1))
PY
rm -rf $out/trace4 $out/pmc4_fetch $out/pmc4_write
