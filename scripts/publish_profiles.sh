#!/bin/bash
# Copies the summaries of gpurun_out/prof_<tag>_* (made by collect_profiles.sh) into profiles/ under the tracked names.
# Usage: bash scripts/publish_profiles.sh r02d
TAG=${1:-r02d}
declare -A NAME=( ["resnet50"]="resnet50_bf16" ["vit_base_patch16_224"]="vit_b16_bf16" ["unicom_ViT_L_14__batch_128"]="unicom_vit_l14_bf16" ["unicom_ViT_L_14__batch_128_dtype_fp8"]="unicom_vit_l14_fp8" )
for k in "${!NAME[@]}"; do
  D=gpurun_out/prof_${TAG}_$k; N=profiles/${TAG}_bench_${NAME[$k]}
  [ -d $D ] || continue
  cp $D/kernel_stats.csv ${N}_kernel_stats.csv; cp $D/last5steps_serialized.csv ${N}_last5steps_serialized.csv
  cp $D/line.json ${N}_line.json; cp $D/line_under_rocprofv3.json ${N}_line_under_rocprofv3.json
  cp $D/pmc_mfma.json ${N}_pmc_mfma.json; cp $D/pmc_traffic.json ${N}_pmc_traffic.json
done
cp gpurun_out/prof_${TAG}_resnet50/pmc_traffic.json profiles/r02_pmc_traffic.json
ls profiles | grep ${TAG}_ | wc -l
