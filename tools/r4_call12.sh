cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
for i in 1 2; do
python bench.py --workload config3 --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('config3 no dist   ', j['ms_per_step'], j['repeat_ms_per_step'])"
python bench.py --workload config3 --force-dist --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('config3 RCCL 1 rk ', j['ms_per_step'], j['repeat_ms_per_step'], j['config']['parallelism'][-40:])"
done
