"""Collates rocprofv3 --pmc counter_collection.csv files: per kernel name, the mean of every counter over its dispatches.
usage: pmc_table.py <dir> <prefix>   (reads <dir>/<prefix>*/**/*counter_collection.csv)"""
import csv, glob, os, sys, collections
root, prefix = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(root, prefix + "*"))):
    if not os.path.isdir(d):
        continue
    grp = os.path.basename(d)[len(prefix):].rsplit("_", 1)[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not ("gemm8p" in k or "wgrad8p" in k or "wgrad256" in k or "conv_igemm" in k):
                continue
            acc[(grp, k.split("(")[0][:48])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (grp, k), ctrs in sorted(acc.items()):
    print(f"== {grp}: {k}")
    for c, v in sorted(ctrs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.0f}   (n={len(v)})")
