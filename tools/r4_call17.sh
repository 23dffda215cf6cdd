cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rank_gpu.py tests/test_xcorrvol_fast_gpu.py tests/test_config4_gpu.py tests/test_photometric_gpu.py tests/test_xcorrvol_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_rank.py 2500 201 2>&1 | tail -1
timeout -k 10 300 python tools/fuzz_volume.py 2000 202 2>&1 | tail -1
timeout -k 10 300 python tools/fuzz_costvol_sep.py 800 203 2>&1 | tail -1
CTD_DS=128,256 timeout -k 10 400 python tools/time_norank.py "" 2>&1 | grep -v amdgpu
./tools/bin/ctd_store_ceiling pattern 54 | grep "TBs"
for m in norank plain; do timeout -k 10 200 python tools/alld_timeline.py tools/variants/libctd_stamps.so $m 2>&1 | grep -E "chunk period|period by|lifetime"; done
