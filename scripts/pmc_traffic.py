"""HBM-side traffic per launch from two rocprofv3 PMC passes over bench.py (FETCH_SIZE and WRITE_SIZE, one pass each,
--kernel-trace only), corrected as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact
for 16-B-per-lane streaming stores and float atomics.  Launches are classified forward / backward by their position
relative to the loss kernel inside each step (steps are delimited by optim_step_kernel launches).
Usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [skip_steps]"""
import csv, json, sys, collections


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    rows.sort()
    return rows


def family(name, phase):
    if "conv_igemm_kernel" in name or "gemm8p_kernel" in name:
        return "conv_igemm_fwd" if phase == "fwd" else "conv_igemm_dgrad"
    for key, fam in (("conv_wgrad_kernel", "conv_wgrad"), ("wgrad3x3_kernel", "conv_wgrad"), ("wgrad8p_kernel", "conv_wgrad"),
                     ("wgrad256_kernel", "conv_wgrad"), ("wgrad_reduce_kernel", "wgrad_reduce"), ("bn_bwd_apply_kernel", "bn_bwd_apply"), ("bn_bwd_reduce_kernel", "bn_bwd_reduce"),
                     ("bn_apply_kernel", "bn_apply")):
        if key in name:
            return fam
    return None


def per_family(rows, skip_steps):
    out = collections.defaultdict(lambda: [0, 0.0])
    step, phase = 0, "fwd"
    for _, name, val in rows:
        if "optim_step_kernel" in name:
            step_end = True
        else:
            step_end = False
        if "loss_fwd" in name or "loss_forward" in name or name.startswith("void loss") or "loss_" in name and phase == "fwd":
            phase = "bwd"
        fam = family(name, phase)
        if fam and step >= skip_steps:
            out[fam][0] += 1
            out[fam][1] += val
        if step_end:
            phase = "fwd"
            nxt = True
        else:
            nxt = False
        if nxt:
            step += 0.5 if False else 0   # (optimizer launches are consecutive per parameter group; count on transition below)
        per_family.last_opt = step_end if not hasattr(per_family, "last_opt") else per_family.last_opt
        if per_family.last_opt and not step_end:
            step += 1
        per_family.last_opt = step_end
    return out


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 1
per_family.last_opt = False
f = per_family(fetch, skip)
per_family.last_opt = False
w = per_family(write, skip)
res = {}
for fam in sorted(set(f) | set(w)):
    nf, kf = f.get(fam, [0, 0.0])
    nw, kw = w.get(fam, [0, 0.0])
    n = max(nf, nw, 1)
    rd, wr = 2.0 * kf * 1024.0 / max(nf, 1), kw * 1024.0 / max(nw, 1)
    res[fam] = dict(launches=n, read_bytes_per_launch=round(rd), write_bytes_per_launch=round(wr), traffic_bytes_per_launch=round(rd + wr))
json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1`",
               correction="bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM section)", kernels=res),
          open(sys.argv[3], "w"), indent=1)
for k, v in res.items():
    print(f"{k:18s} n={v['launches']:5d} read {v['read_bytes_per_launch']/1e6:8.1f} MB  write {v['write_bytes_per_launch']/1e6:8.1f} MB per launch")
