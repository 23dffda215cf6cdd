cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do export CTD_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libctd_$v.so
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv_$v -- python tools/resolve_probe.py 1e-5 > /dev/null 2>&1; f=$(ls -t gpurun_out/pv_$v/*/*kernel_stats.csv | head -1); python - $f $v <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "argmax" in r["Name"] or "t256" in r["Name"]: print("  ", sys.argv[2], r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
