#!/bin/bash
# Runs on the GPU box: the four bench lines of the round (ms per step, whole-job throughput) without baselines / roofline legs.
# Usage: bash scripts/bench_all.sh [steps] [warmup]
S=${1:-30}; W=${2:-8}
run() { python bench.py "$@" --steps $S --warmup $W --no-cpu-baseline --no-host-work --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['dtype'], d['ms_per_step'], d['value'])"; }
run --model resnet50 && run --model vit_base_patch16_224 && run --model "unicom ViT-L/14" --batch 128 && run --model "unicom ViT-L/14" --batch 128 --dtype fp8
