cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4 0" "4 23000" "2 0" "2 23000" "1 0" "1 23000" "4 27000"; do set -- $cfg; export CTD_SCAN_PX=$1 CTD_SCAN_LDS=$2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof10_$1_$2 -- python tools/resolve_probe.py 1e-5 > /dev/null 2>&1; f=$(ls -t gpurun_out/prof10_$1_$2/*/*kernel_stats.csv | head -1); python - $f $1 $2 <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "argmax_scan" in r["Name"]: print("  px", sys.argv[2], "lds", sys.argv[3], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
