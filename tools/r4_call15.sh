cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
bash tools/profile_round.sh round4 > gpurun_out/r4n/profile_round.log 2>&1; echo "profile_round rc $?"
grep timed gpurun_out/r4n/profile_round.log
bash tools/profile_cfg4.sh round4 > gpurun_out/r4n/profile_cfg4.log 2>&1; echo "profile_cfg4 rc $?"
grep "calls" gpurun_out/r4n/profile_cfg4.log | head -8
bash tools/profile_cfg3.sh > gpurun_out/r4n/profile_cfg3.log 2>&1; echo "profile_cfg3 rc $?"
tail -12 gpurun_out/r4n/profile_cfg3.log
python bench.py > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err; echo "bench rc $?"
python -c "
import json; j=json.load(open('gpurun_out/r4n/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['disparity_mae_vs_ref']); print(j['roofline']['store_only_ceiling']); print(j['also_measured']['fused_volume_free']['ms_per_step'], j['also_measured']['volume_kernel_alone']['avg_launch_ms'])"
python bench.py --workload config4 > gpurun_out/r4n/bench_cfg4.json 2>/dev/null; echo "cfg4 rc $?"
