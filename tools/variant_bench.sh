cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do export CTD_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libctd_$v.so
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pv_$v -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; f=$(ls -t gpurun_out/pv_$v/*/*kernel_stats.csv | head -1); python - $f $v <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in sys.argv[3:] or ["fixup","t256","scan","resolve"]): print("  ", sys.argv[2], r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
