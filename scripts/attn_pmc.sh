#!/bin/bash
# GPU box: PMC counters of the fused attention kernels (standalone microbenchmark), one counter group per pass.
# Usage: bash scripts/attn_pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]   ->  gpurun_out/attn_pmc_<tag>/
TAG=${1:-a}; shift; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/attn_pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/scripts/attn_microbench.py > $OUT/p$i.log 2>&1 && echo "pass $i ok"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        k = k[k.index("attn"):][:24]
        acc[(k, r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
rm -rf $OUT/p*/*/*.db $OUT/p*/pmc_counter_collection.csv
