#!/bin/bash
# Copies the summaries of gpurun_out/prof_<tag>_* (scripts/collect_profiles.sh) into profiles/ under the names bench.py and the docs use.
# Usage: bash scripts/publish_profiles.sh r03
TAG=${1:-r03}
declare -A NAME=( ["resnet50"]="resnet50_bf16" ["vit_base_patch16_224"]="vit_b16_bf16" ["unicom_ViT_L_14__batch_128"]="unicom_vit_l14_bf16" ["unicom_ViT_L_14__batch_128_dtype_fp8"]="unicom_vit_l14_fp8" )
for k in "${!NAME[@]}"; do
  D=gpurun_out/prof_${TAG}_$k; N=${NAME[$k]}
  [ -d "$D" ] || continue
  cp $D/line.json profiles/${TAG}_bench_${N}_line.json
  cp $D/line_under_rocprofv3.json profiles/${TAG}_bench_${N}_line_under_rocprofv3.json
  cp $D/kernel_stats.csv profiles/${TAG}_bench_${N}_kernel_stats.csv
  cp $D/last5steps_serialized.csv profiles/${TAG}_bench_${N}_last5steps_serialized.csv
  cp $D/pmc_mfma.json profiles/${TAG}_bench_${N}_pmc_mfma.json
  cp $D/pmc_traffic.json profiles/${TAG}_${N}_pmc_traffic.json
  [ -f $D/overlap.txt ] && cp $D/overlap.txt profiles/${TAG}_bench_${N}_overlap.txt
  echo "published $N"
done
