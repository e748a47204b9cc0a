#!/bin/bash
# GPU box: kernel trace of the default bench -> scripts/idle_gaps.py (where the GPU idles inside an overlapped step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/idle; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-work --steps 12 --warmup 6 "$@" > $OUT/line.json 2> $OUT/trace.err
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/idle_gaps.py $T 5 > $OUT/idle_gaps.txt
rm -rf $OUT/trace
cat $OUT/idle_gaps.txt
