cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NKB_GRAM_MAX_C=256 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gram256 -o t -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-host-work --no-roofline > $R/gpurun_out/gram256.json 2>/dev/null
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/gram256/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "gram" in n or "gemm_tn" in n or "wgrad_reduce" in n or "bn_apply" in n: print(n[:70], r["Calls"], r["AverageNs"], r["TotalDurationNs"])
PY
cat $R/gpurun_out/gram256.json | python3 -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"
