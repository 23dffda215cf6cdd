cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
python -m pytest tests/test_photometric_gpu.py tests/test_config4_gpu.py tests/test_xcorrvol_fast_gpu.py tests/test_rank_gpu.py -x -q 2>&1 | tail -5
timeout -k 10 300 python tools/time_costvol.py > gpurun_out/r4d/costvol.txt 2>&1
cat gpurun_out/r4d/costvol.txt
