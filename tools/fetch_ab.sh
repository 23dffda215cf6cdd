#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the all-D kernel per library variant: tools/fetch_ab.sh out_dir lib.so [lib.so ...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
for lib in "$@"; do
  n=$(basename $lib .so)
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$n.f -- python tools/prof_run.py rank 16 $lib > $out/$n.f.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/$n.w -- python tools/prof_run.py rank 16 $lib > $out/$n.w.log 2>&1
done
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*.[fw]")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "alld" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(d.split("/")[-1], {k: round(sum(v) / len(v) / 1024.0, 1) for k, v in agg.items()}, "MiB per launch (FETCH_SIZE under-reports 2x)")
PY
