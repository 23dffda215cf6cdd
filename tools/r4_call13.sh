cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
python -m pytest tests/test_rccl_one_rank_gpu.py tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -3
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r4m/bench_cfg4.json 2>gpurun_out/r4m/cfg4.err; echo rc $?; python -c "
import json; j=json.load(open('gpurun_out/r4m/bench_cfg4.json')); print(j['ms_per_step'], j['roofline']['frac'], j['ncc_two_frames_per_call'], j['disparity_mae_vs_ref'])"
