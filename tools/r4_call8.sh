cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout -k 10 400 python tools/fuzz_costvol_sep.py 600 5 > gpurun_out/r4h/fuzz_costvol_sep.txt 2>&1; tail -4 gpurun_out/r4h/fuzz_costvol_sep.txt
timeout -k 10 300 python tools/fuzz_photometric.py 600 6 > gpurun_out/r4h/fuzz_photometric.txt 2>&1; tail -2 gpurun_out/r4h/fuzz_photometric.txt
timeout -k 10 300 python tools/fuzz_volume.py 1500 43 > gpurun_out/r4h/fuzz_volume.txt 2>&1; tail -2 gpurun_out/r4h/fuzz_volume.txt
timeout -k 10 300 python tools/fuzz_losses.py 300 7 > gpurun_out/r4h/fuzz_losses.txt 2>&1; tail -2 gpurun_out/r4h/fuzz_losses.txt
