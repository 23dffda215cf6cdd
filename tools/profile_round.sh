#!/bin/bash
# Collects the judged profile artefacts of one round into gpurun_out/profiles_<tag>/ (copy to profiles/ afterwards).
tag=${1:-round1}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. kernel trace + stats of the exact bench command (no PMC in this pass)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1 || exit 1
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
# 2. HBM traffic counters, separate passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $out/pmc_write -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summ = {}
for k, cs in agg.items():
    if "ctd::" in k:
        summ[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        summ[k]["launches_sampled"] = max(len(v) for v in cs.values())
json.dump(summ, open("%s/%s_pmc_hbm_traffic.json" % (out, tag), "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
