cd $GRAFT_REPO_ROOT
for v in "$@"; do export CTD_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libctd_$v.so
echo "$v: $(python bench.py --steps 20 --warmup 3 --no-cpu-baseline | grep -o '"avg_launch_ms": [0-9.]*\|"disparity_mae_vs_ref": [0-9.]*' | tr '\n' ' ')"; done
