#!/bin/bash
# usage: pmc_one_convp.sh <tag> <N> <H> <Cin> <Cout> <kind>   (run on the GPU box; writes gpurun_out/pmcp_<tag>_<pass>/)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
            "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_${tag}_$i -o p -- python3 $R/scripts/one_convp.py "$@" > $R/gpurun_out/pmcp_${tag}_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 6):
    for f in glob.glob("$R/gpurun_out/pmcp_${tag}_%d/**/p_counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "convp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("%-32s %14.0f  (n=%d)" % (k, sum(v[-3:]) / max(1, len(v[-3:])), len(v)))
PY
