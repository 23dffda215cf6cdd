set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python -m pytest tests/test_rank_gpu.py tests/test_xcorrvol_fast_gpu.py tests/test_config4_gpu.py tests/test_xcorrvol_gpu.py -x -q > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4b/pytest.log
tail -5 gpurun_out/r4b/pytest.log
timeout -k 10 300 python tools/time_variant.py tools/variants/libctd_base.so "" tools/variants/libctd_ab1.so tools/variants/libctd_ab2.so tools/variants/libctd_ab3.so > gpurun_out/r4b/time_variant.txt 2>&1
cat gpurun_out/r4b/time_variant.txt
for m in rank norank; do timeout -k 10 200 python tools/alld_timeline.py tools/variants/libctd_stamps.so $m > gpurun_out/r4b/timeline_$m.txt 2>&1; done
cat gpurun_out/r4b/timeline_rank.txt
