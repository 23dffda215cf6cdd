cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 400 python tools/time_norank.py "" tools/variants/libctd_ab4.so tools/variants/libctd_ab3.so > gpurun_out/r4f/time_ab4.txt 2>&1
cat gpurun_out/r4f/time_ab4.txt
./tools/bin/ctd_store_ceiling pattern 54
