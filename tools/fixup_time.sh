#!/bin/bash
# fix-up / resolve kernel times by call type (ranked vs plain) from a rocprofv3 trace of tools/time_variant.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fixprof
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fixprof -- python tools/time_variant.py > gpurun_out/fixprof.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fixprof/*/*kernel_trace.csv")[0]
prev, acc = None, collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    for key in ("fixup_kernel", "resolve", "merge", "prepass", "runs"):
        if key in n:
            acc[(key, prev[-40:] if key == "fixup_kernel" and prev else "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    prev = n
for k, v in sorted(acc.items()):
    print("%-14s %-42s n=%3d avg %6.1f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
PY
grep argmax gpurun_out/fixprof.log | head -2
