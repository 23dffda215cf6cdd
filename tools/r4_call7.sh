cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 500 python tools/fuzz_rank.py 1500 41 > gpurun_out/r4g/fuzz_rank.txt 2>&1; tail -3 gpurun_out/r4g/fuzz_rank.txt
timeout -k 10 300 python tools/fuzz_rank.py 150 42 big > gpurun_out/r4g/fuzz_rank_big.txt 2>&1; tail -3 gpurun_out/r4g/fuzz_rank_big.txt
timeout -k 10 400 python tools/fuzz_volume.py 1500 43 > gpurun_out/r4g/fuzz_volume.txt 2>&1; tail -3 gpurun_out/r4g/fuzz_volume.txt
bash tools/fetch_ab.sh gpurun_out/r4g/fetch connecting_the_dots_amd/libctd_hip.so tools/variants/libctd_ab3.so 2>&1 | tail -6
