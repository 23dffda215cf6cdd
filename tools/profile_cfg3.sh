cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3/trace -- python bench.py --workload config3 --headline-only --no-cpu-baseline > gpurun_out/c3/bench.log 2>&1 || exit 1
grep '^{"metric"' gpurun_out/c3/bench.log | cut -c1-400
python tools/kstats.py gpurun_out/c3/trace | head -20
