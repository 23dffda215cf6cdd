cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python -m pytest tests/test_rank_gpu.py tests/test_xcorrvol_fast_gpu.py tests/test_config4_gpu.py -x -q 2>&1 | tail -2
CTD_DS=128,256 timeout -k 10 400 python tools/time_norank.py tools/variants/libctd_base.so "" > gpurun_out/r4c/time_norank.txt 2>&1
cat gpurun_out/r4c/time_norank.txt
for m in norank plain; do timeout -k 10 200 python tools/alld_timeline.py tools/variants/libctd_stamps.so $m > gpurun_out/r4c/timeline_$m.txt 2>&1; done
cat gpurun_out/r4c/timeline_norank.txt gpurun_out/r4c/timeline_plain.txt
