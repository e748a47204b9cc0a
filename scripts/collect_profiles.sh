#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace + stats of the default bench command, the PMC passes the
# roofline's `traffic` and MFMA-utilisation figures come from, and the bench lines themselves; results under gpurun_out/prof_$TAG,
# summaries are copied to profiles/ by hand afterwards (profiles/ is tracked, gpurun_out/ is scratch).
# Usage: bash scripts/collect_profiles.sh <tag> [model] [extra bench.py arguments, e.g. --batch 128 --dtype fp8]
set -o pipefail
TAG=${1:-r03}; MODEL=${2:-resnet50}; shift; shift; EXTRA="$*"; SUFFIX=$(echo "$EXTRA" | tr -c 'a-zA-Z0-9' '_' | sed 's/__*/_/g; s/_$//')
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${MODEL//[^a-zA-Z0-9]/_}${SUFFIX:+_$SUFFIX}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-host-work $EXTRA --model"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $B "$MODEL" --steps 20 --warmup 8 > $OUT/line_under_rocprofv3.json 2> $OUT/trace.err && echo "trace ok"
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $B "$MODEL" --steps 2 --warmup 1 --no-roofline > /dev/null 2> $OUT/pmc_$C.err && echo "pmc $C ok"
done
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 scripts/profile_summary.py $T 5 $OUT/last5steps_serialized.csv > /dev/null && cp $S $OUT/kernel_stats.csv
F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W $OUT/pmc_traffic.json 1 > $OUT/pmc_traffic.txt
# the bench line LAST, complete: its roofline.traffic comes from the PMC passes just made (bench.py takes the newest
# profiles/rNN_<workload>_pmc_traffic.json), and it carries cpu_baseline (every committed line has both blocks)
declare -A PUB=( ["resnet50"]="resnet50_bf16" ["vit_base_patch16_224"]="vit_b16_bf16" ["unicom ViT-L/14--batch 128"]="unicom_vit_l14_bf16" ["unicom ViT-L/14--batch 128 --dtype fp8"]="unicom_vit_l14_fp8" )
N=${PUB["$MODEL${EXTRA:+--$EXTRA}"]}
[ -n "$N" ] && cp $OUT/pmc_traffic.json profiles/${TAG}_${N}_pmc_traffic.json
python3 $ROOT/bench.py $EXTRA --model "$MODEL" --steps 30 --warmup 8 > $OUT/line.json 2> $OUT/line.err && echo "line ok"
python3 scripts/overlap_report.py $T 5 4 > $OUT/overlap.txt
M=$(find $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1); MT=$(find $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*kernel_trace.csv" | head -1)
python3 scripts/pmc_mfma.py $M $MT $OUT/pmc_mfma.json > $OUT/pmc_mfma.txt
rm -rf $OUT/trace/*/*.db $OUT/pmc_*/*/*.db 2>/dev/null
du -sh $OUT; cat $OUT/line.json | head -c 600; echo; cat $OUT/pmc_mfma.txt; head -30 $OUT/pmc_traffic.txt
[ -s $OUT/line.json ] && [ -s $OUT/kernel_stats.csv ]
