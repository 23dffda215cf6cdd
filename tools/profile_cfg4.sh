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
rm -rf $out/trace4
