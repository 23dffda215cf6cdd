# rocprofv3 kernel statistics of tools/time_fused.py for the given modes; usage: bash tools/r5_prof_fused.sh OUTDIR MODE...
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/prof -- python3 $GRAFT_REPO_ROOT/tools/time_fused.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py gpurun_out/$out/prof | grep -v "at::\|elementwise\|Memset\|fill" 
