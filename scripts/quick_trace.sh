#!/bin/bash
# GPU box: kernel-trace of the default bench (short) -> serialized per-kernel table of the last 5 (HIP-event profiled) steps.
# Usage: bash scripts/quick_trace.sh <tag> [bench args...]   (environment switches are inherited)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/qt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-work --steps 12 --warmup 6 "$@" > $OUT/line.json 2> $OUT/trace.err
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/profile_summary.py $T 5 $OUT/last5.csv
python3 scripts/overlap_report.py $T 5 4 > $OUT/overlap.txt
python3 scripts/step_dump.py $T > $OUT/step_dump.txt
rm -rf $OUT/trace
head -45 $OUT/last5.csv
