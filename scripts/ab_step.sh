# A/B of whole train steps with alternative library builds on one box: scripts/ab_step.sh <rounds> "<bench.py arguments>" <name> [<name> ...]
# (name: default | an alt build of scripts/build_alt.sh); prints ms_per_step per run
set -o pipefail
O=gpurun_out/ab_step; mkdir -p $O; R=$1; ARGS=$2; shift 2
for r in $(seq 1 $R); do
 for v in "$@"; do
  if [ $v = default ]; then unset NKBHIP_LIB; else export NKBHIP_LIB=$PWD/build/alt_$v/libnkbhip.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-work --no-roofline $ARGS > $O/${v}_$r.json 2> $O/${v}_$r.err || { tail -n 5 $O/${v}_$r.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("$O/${v}_$r.json").read().strip().splitlines()[-1]); print("$v", $r, d["ms_per_step"], d["value"])
PY
 done
done
