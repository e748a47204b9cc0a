#!/bin/bash
# GPU box: rocprofv3 kernel trace of scripts/wr_check.py time (or w3_check.py) -> per-kernel average durations.  usage: wr_trace.sh <tag> <script> [env assignments are inherited]
TAG=$1; SCR=${2:-scripts/wr_check.py}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/$SCR time > $OUT/out.txt 2> $OUT/trace.err
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/summarize_trace.py $T 1 30 > $OUT/summary.txt
rm -rf $OUT/trace
cat $OUT/summary.txt
