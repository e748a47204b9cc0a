#!/bin/bash
# usage: pmc_one.sh <tag> <kernel-name-substring> <script.py> [args ...]   (GPU box; SQ counter passes over one small program)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
tag=$1; kname=$2; prog=$3; shift 3
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc1_${tag}_$i -o p -- python3 $R/scripts/$prog "$@" > $R/gpurun_out/pmc1_${tag}_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 4):
    for f in glob.glob("$R/gpurun_out/pmc1_${tag}_%d/**/p_counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$kname" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("%-32s %14.0f  (n=%d)" % (k, sum(v[-3:]) / max(1, len(v[-3:])), len(v)))
PY
