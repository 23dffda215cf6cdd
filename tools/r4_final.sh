cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4final
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile_round.sh round4 > gpurun_out/r4final/profile_round.log 2>&1; echo "profile_round rc $?"
grep timed gpurun_out/r4final/profile_round.log
python -c "
import json; j=json.loads(open('gpurun_out/profiles_round4/round4_bench_under_rocprof.json').read()); print('traced:', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
python bench.py > gpurun_out/r4final/bench.json 2> gpurun_out/r4final/bench.err; echo "bench rc $?"
python -c "
import json; j=json.load(open('gpurun_out/r4final/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['disparity_mae_vs_ref']); print(j['roofline']['store_only_ceiling']['GBs'], j['roofline']['store_only_ceiling']['card_best_GBs'])"
timeout -k 10 200 python tools/fuzz_rank.py 1500 7 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_volume.py 1500 8 2>&1 | tail -1
