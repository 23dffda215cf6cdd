#!/bin/bash
# tools/ab_lcn.sh LIB...  -- rocprofv3 kernel time of the LCN kernel per library variant (tools/prof_lcn.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  n=$(basename $lib .so)
  rm -rf gpurun_out/abl_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$n -- python tools/prof_lcn.py $lib > gpurun_out/abl_$n.log 2>&1
  echo "== $n"; python tools/kstats.py gpurun_out/abl_$n | grep -i lcn
done
