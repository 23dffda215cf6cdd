cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 300 python tools/fuzz_rank.py 1200 51 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_volume.py 1200 52 2>&1 | tail -2
timeout -k 10 200 python tools/fuzz_fast.py 2>&1 | tail -3
python bench.py --no-cpu-baseline > gpurun_out/r4j/bench.json 2>gpurun_out/r4j/bench.err; python -c "
import json; j=json.load(open('gpurun_out/r4j/bench.json')); print(j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['disparity_mae_vs_ref'], j['roofline']['store_only_ceiling'])"
