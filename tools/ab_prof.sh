#!/bin/bash
# tools/ab_prof.sh LIB...  -- rocprofv3 kernel stats of tools/time_variant.py for each experimental library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  n=$(basename $lib .so)
  rm -rf gpurun_out/ab_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$n -- python tools/time_variant.py $lib > gpurun_out/ab_$n.log 2>&1
  echo "== $n"; python tools/kstats.py gpurun_out/ab_$n | head -9
done
