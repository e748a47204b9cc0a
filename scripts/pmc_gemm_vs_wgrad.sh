#!/bin/bash
# usage (on the GPU box): pmc_gemm_vs_wgrad.sh <tag>
# SQ counters of one ViT-B/16-shaped launch (M ~ 50 k tokens, 768 <-> 3072) of the eight-phase forward GEMM and of the
# eight-phase weight gradient; results under gpurun_out/pmcgw_<tag>_<kernel>_<pass>/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
tag=$1
for which in gemm wgrad; do
  if [ $which = gemm ]; then prog="$R/scripts/one_conv.py 768 3072 1 1 14 0"; else prog="$R/scripts/one_wgrad.py"; fi
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
              "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmcgw_${tag}_${which}_$i -o p -- python3 $prog > $R/gpurun_out/pmcgw_${tag}_${which}_$i.log 2>&1 || { tail -n 5 $R/gpurun_out/pmcgw_${tag}_${which}_$i.log; exit 1; }
  done
done
python3 $R/scripts/pmc_table.py $R/gpurun_out "pmcgw_${tag}_" | tee $R/gpurun_out/pmcgw_${tag}.txt
