# One-rank rehearsal of the multi-GPU step (the reducer's kernels and stream choreography, no wire time): plain | fp32 buckets | bf16 buckets.
# Usage (GPU box): bash scripts/rehearse.sh <tag>      -> gpurun_out/<tag>_rehearse_<model>_<mode>.json
TAG=${1:-r05}
for a in "resnet50|" "unicom ViT-L/14|--batch 128"; do
  m="${a%%|*}"; e="${a##*|}"; n=$(echo "$m" | tr 'A-Z' 'a-z' | tr -c 'a-z0-9' '_' | sed 's/__*/_/g; s/_$//'); [ "$n" = unicom_vit_l_14 ] && n=unicom_vit_l14
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-work --no-roofline --model "$m" $e --steps 20 --warmup 6 2>/dev/null | tail -n 1 > gpurun_out/${TAG}_rehearse_${n}_plain.json
  for mode in fp32 bf16; do
    NKB_FORCE_REDUCER=1 NKB_GRAD_BUCKET_DTYPE=$mode timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-work --no-roofline --model "$m" $e --steps 20 --warmup 6 2>/dev/null | tail -n 1 > gpurun_out/${TAG}_rehearse_${n}_$mode.json
  done
  for mode in plain fp32 bf16; do python -c "import json; d=json.load(open('gpurun_out/${TAG}_rehearse_${n}_$mode.json')); print('$n $mode', d['ms_per_step'])"; done
done
