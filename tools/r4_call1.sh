set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/pytest.log
tail -5 gpurun_out/r4a/pytest.log
timeout -k 10 300 ./tools/bin/store_streams > gpurun_out/r4a/store_streams.txt 2>&1
tail -3 gpurun_out/r4a/store_streams.txt
timeout -k 10 300 python tools/d_sweep.py > gpurun_out/r4a/d_sweep.txt 2>&1
tail -16 gpurun_out/r4a/d_sweep.txt
for m in rank norank plain; do timeout -k 10 200 python tools/alld_timeline.py tools/variants/libctd_stamps.so $m > gpurun_out/r4a/timeline_$m.txt 2>&1; done
cat gpurun_out/r4a/timeline_rank.txt
