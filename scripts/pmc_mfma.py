"""MFMA utilisation per kernel family from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, --kernel-trace only) over
bench.py: utilisation = sum of MFMA-busy cycles over the launches / (sum of launch durations x 2.4 GHz x 1024 SIMDs).
Usage: python scripts/pmc_mfma.py <counter_collection.csv> <kernel_trace.csv> <out.json>"""
import csv, collections, json, sys
CLOCK_GHZ, SIMDS = 2.4, 256 * 4
ctr, name, dur = collections.defaultdict(float), {}, {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        d = int(r["Dispatch_Id"]); ctr[d] += float(r["Counter_Value"])
        name[d] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
for r in csv.DictReader(open(sys.argv[2])):
    dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
fam = collections.defaultdict(lambda: [0, 0.0, 0])
for d, busy in ctr.items():
    n = name[d]
    key = next((k for k in ("conv_igemm", "convp", "conv1p", "conv1s", "stempw", "stemp", "gramr", "gemm8p", "conv_wgrad", "wgrad3x3", "wgradr", "wgrad8p", "wgrad256", "attn_fwd", "attn_bwd") if k in n), None)
    if key and d in dur:
        f = fam[key]; f[0] += 1; f[1] += busy; f[2] += dur[d]
out = {k: dict(launches=f[0], mfma_busy_cycles=f[1], duration_ns=f[2],
               mfma_utilisation=round(f[1] / (f[2] * CLOCK_GHZ * SIMDS), 4)) for k, f in fam.items()}
json.dump(dict(clock_ghz=CLOCK_GHZ, simds=SIMDS, kernels=out), open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:12s} n={v['launches']:5d} MFMA busy {100 * v['mfma_utilisation']:.1f} % of {SIMDS} SIMDs")
