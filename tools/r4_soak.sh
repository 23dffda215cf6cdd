cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4soak
timeout -k 10 500 python tools/fuzz_rank.py 12000 101 > gpurun_out/r4soak/fuzz_rank.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_rank.txt
timeout -k 10 500 python tools/fuzz_rank.py 900 102 big > gpurun_out/r4soak/fuzz_rank_big.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_rank_big.txt
timeout -k 10 400 python tools/fuzz_volume.py 8000 103 > gpurun_out/r4soak/fuzz_volume.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_volume.txt
timeout -k 10 400 python tools/fuzz_costvol_sep.py 4000 104 > gpurun_out/r4soak/fuzz_costvol_sep.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_costvol_sep.txt
timeout -k 10 400 python tools/fuzz_photometric.py 2000 105 > gpurun_out/r4soak/fuzz_photometric.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_photometric.txt
timeout -k 10 300 python tools/fuzz_lcn.py 1500 106 > gpurun_out/r4soak/fuzz_lcn.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_lcn.txt
timeout -k 10 300 python tools/fuzz_pattern_loss.py 400 107 > gpurun_out/r4soak/fuzz_pattern_loss.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_pattern_loss.txt
timeout -k 10 300 python tools/fuzz_nn.py > gpurun_out/r4soak/fuzz_nn.txt 2>&1; tail -1 gpurun_out/r4soak/fuzz_nn.txt
grep -h "^case" gpurun_out/r4soak/*.txt | head -20
