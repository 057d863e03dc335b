#!/bin/bash
# usage: tools/lds_conflict_pass.sh <outdir> <config> name=lib.so [name=lib.so ...]
# One rocprofv3 --pmc pass (LDS counters only) of tools/prof_fwdbwd.py per library build; prints the
# fused kernel's SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per launch for each.
out=$1; cfg=$2; shift 2
export TMPDIR=/tmp
mkdir -p $out
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  PPO_HIP_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS \
      --output-format csv -d $out/$name -- python3 tools/prof_fwdbwd.py $cfg 3 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 1; }
done
python3 - "$out" "$cfg" <<'PY'
import csv, glob, os, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(out + "/*/")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith("void fwdbwd_kernel"):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    print(f"{cfg} {os.path.basename(d[:-1]):10s} " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(a.items())) +
          f"  conflict/active={a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.4f}")
PY
