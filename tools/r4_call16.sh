cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
bash tools/profile_round.sh round4 > gpurun_out/r4o/profile_round.log 2>&1; echo "profile_round rc $?"
grep timed gpurun_out/r4o/profile_round.log
python -c "
import json; j=json.loads(open('gpurun_out/profiles_round4/round4_bench_under_rocprof.json').read()); print(j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
