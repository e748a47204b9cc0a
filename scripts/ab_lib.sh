#!/bin/bash
# usage: ab_lib.sh <alt .so path relative to the repo> [rounds] [bench args]  — alternating bench.py runs on the built library and on an
# alternative build of it (NKBHIP_LIB); prints ms/step per run
ALT=$1; ROUNDS=${2:-2}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq $ROUNDS); do
  for v in default alt; do
    if [ $v = alt ]; then export NKBHIP_LIB=$R/$ALT; else unset NKBHIP_LIB; fi
    python bench.py --steps 30 --warmup 10 --no-host-work --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], 'loss', d['final_loss'])"
  done
done
