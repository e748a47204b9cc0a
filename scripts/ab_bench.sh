#!/bin/bash
# usage: ab_bench.sh VAR "v1 v2 ..." [rounds] [bench args]   — alternating runs of bench.py under VAR=value on one box; prints ms/step per value
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 30 --warmup 10 --no-host-work --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']
print('$VAR=$v', d['ms_per_step'], 'loss', d['final_loss'], 'bn_fin', k.get('bn_finalize'), 'bn_bwd_red', k.get('bn_bwd_reduce'))"
  done
done
