#!/bin/bash
# usage: tools_pmc.sh <outdir> <algo>   -- one rocprofv3 --pmc pass per counter group (no tracing domains mixed in)
out=$1; algo=${2:-fast}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python tools/prof_run.py $algo > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.p$i.log; }
done <<'GRP'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
GRP
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "_summary.txt", "w") as fo:
    for k, cs in agg.items():
        if "ctd::" not in k: continue
        fo.write("== %s\n" % k)
        for c, v in sorted(cs.items()):
            fo.write("  %-40s n=%d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "_summary.txt").read())
PY
