#!/bin/bash
# GPU box: same-box A/B of one environment switch on one bench command, alternating values (box-to-box spread is 2-3 %).
# Usage: bash scripts/ab_env.sh VAR "v1 v2 v1 v2" <bench.py arguments...>
VAR=$1; VALS=$2; shift; shift
for v in $VALS; do
  env $VAR=$v python bench.py "$@" --steps 30 --warmup 8 --no-cpu-baseline --no-host-work --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'])"
done
