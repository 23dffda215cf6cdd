cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
bash tools/profile_round.sh round4b > gpurun_out/r4k/profile_round.log 2>&1; echo "profile_round rc $?"
grep timed gpurun_out/r4k/profile_round.log
python bench.py > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err; echo "bench rc $?"
python -c "
import json; j=json.load(open('gpurun_out/r4k/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['disparity_mae_vs_ref']); print(j['roofline']['store_only_ceiling']); print(j['also_measured']['fused_volume_free']['ms_per_step'], j['also_measured']['volume_kernel_alone']['avg_launch_ms'])"
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r4k/bench_cfg4.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r4k/bench_cfg4.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['census']['avg_launch_ms'], j['sad']['avg_call_ms'], j['disparity_mae_vs_ref'])"
timeout -k 10 200 python tools/fuzz_lcn.py 2>&1 | tail -2
