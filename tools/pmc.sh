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
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
GRBM_GUI_ACTIVE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
FETCH_SIZE
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
