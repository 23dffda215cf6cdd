cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4e/pytest.log
tail -3 gpurun_out/r4e/pytest.log
python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; echo "bench rc $?"
cut -c1-1500 gpurun_out/r4e/bench.json
bash tools/profile_round.sh round4 > gpurun_out/r4e/profile_round.log 2>&1; echo "profile_round rc $?"
tail -12 gpurun_out/r4e/profile_round.log
bash tools/profile_cfg4.sh round4 > gpurun_out/r4e/profile_cfg4.log 2>&1; echo "profile_cfg4 rc $?"
tail -30 gpurun_out/r4e/profile_cfg4.log
bash tools/pmc.sh gpurun_out/r4e/sq_norank norank > gpurun_out/r4e/pmc_norank.log 2>&1
bash tools/pmc.sh gpurun_out/r4e/sq_rank rank > gpurun_out/r4e/pmc_rank.log 2>&1
grep -A40 "alld" gpurun_out/r4e/sq_norank_summary.txt | head -50
