#!/bin/bash
# tools/ab_tail.sh NAME...: rocprofv3 kernel times of the ranked call's side kernels, in-tree build and tools/variants/libctd_NAME.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tail
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib=tools/variants/libctd_$v.so
  rm -rf gpurun_out/tail/x$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tail/x$v -- python ${CTD_AB_TARGET:-tools/time_tail.py} $lib > gpurun_out/tail/log$v.txt 2>&1 || exit 1
  echo "== ${v:-in-tree}"
  python3 - gpurun_out/tail/x$v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0]
    if any(k in n for k in ("tail","fixup","prepass","alld","lcn","geometric","d2d","finish")): print("  %-40s calls %s avg %.1f us min %.1f" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
