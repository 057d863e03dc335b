#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <config> <epochs>   (one rocprofv3 --pmc run per counter group)
out=$1; cfg=${2:-c4}; ep=${3:-3}
export TMPDIR=/tmp
mkdir -p $out
i=0
while read -r group; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $group --output-format csv -d $out/p$i -- python3 tools/prof_fwdbwd.py $cfg $ep > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32
SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_CYCLES SQ_WAVES
GRBM_GUI_ACTIVE FETCH_SIZE
GRBM_GUI_ACTIVE WRITE_SIZE
GROUPS
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"   {c:32s} n={len(v):4d} avg={sum(v)/len(v):.6g}\n")
print(open(out + "/summary.txt").read())
PY
