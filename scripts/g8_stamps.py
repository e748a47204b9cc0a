"""Where a gemm8p tile spends its time (diagnostic build: scripts/build_alt.sh stamps "-DNKB_G8_STAMPS" gemm8p.hip, run with
NKBHIP_LIB=build/alt_stamps/libnkbhip.so): in-kernel s_memtime stamps of wave 0 / wave 4 at the top of a tile's first four k-tiles,
in front of and behind its epilogue.  python scripts/g8_stamps.py M K N [epi]   (epi: plain | bias | add | aux | gelu)"""
import ctypes, os, sys, statistics, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
M, K, N = [int(v) for v in sys.argv[1:4]]
epi = sys.argv[4] if len(sys.argv) > 4 else "plain"
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.03).to(T); b = torch.zeros(N, device=dev)
y = torch.empty(M, N, device=dev, dtype=T); y2 = torch.empty(M, N, device=dev, dtype=T); aux = torch.randn(M, N, device=dev).to(T)
def run():
    if epi == "gelu": hip.linear_gelu(d, 5, x, w, b, None, y, y2, M, K, N)
    elif epi == "aux": hip.linear_gelu(d, 4, x, w, None, aux, y, None, M, K, N)
    else:
        hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N, bias=b if epi in ("bias", "add") else None,
                      add=aux if epi == "add" else None, ldadd=N if epi == "add" else 0)
hip.gemm8p_config(True, 1, 128)
for _ in range(3): run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); e.record(); torch.cuda.synchronize()
us = a.elapsed_time(e) * 1e3
lib = hip.load()
n = 256 * 12 * 8 * 2
buf = (ctypes.c_ulonglong * n)()
assert lib.nkb_g8_read_stamps(buf) == 0
import numpy as np
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 12, 8, 2).astype(np.int64)
tiles = (M + 255) // 256 * (N // 256)
KT = K // 64
written = (st[:, :, 7, 0] > 0)
grid = int(written[:, 0].sum()); per_wg = written.sum(1)
rounds = int(per_wg.max())
full = per_wg == rounds
# s_memtime is per XCD (not synchronised across dies): only differences inside one workgroup mean anything
span = np.array([st[w, per_wg[w] - 1, 7, :].max() - st[w, 0, 0, :].min() for w in range(256) if per_wg[w] > 0 and full[w]])
tpu = float(np.median(span)) / us          # ticks per microsecond, taking the full-length workgroups' span as the kernel's duration
print(f"M={M} K={K} N={N} epi={epi}: {us:.1f} us, {grid} workgroups, {int(full.sum())} of them walk {rounds} tiles, k-tiles/tile {KT}; ~{tpu:.0f} ticks/us")
def med(a): return float(np.median(a)) / tpu
for wv in (0, 1):
    print(f" wave group {wv}:")
    for r in range(rounds):
        s = st[full, r, :, wv]
        kt = [med(s[:, i + 1] - s[:, i]) for i in range(3)]
        main = med(s[:, 4] - s[:, 0]); bar = med(s[:, 5] - s[:, 4]); ep = med(s[:, 6] - s[:, 5]); bar2 = med(s[:, 7] - s[:, 6])
        steady = med(s[:, 4] - s[:, 3]) / max(1, KT - 3)
        nxt = med(st[full, r + 1, 0, wv] - s[:, 7]) if r + 1 < rounds else float("nan")
        print(f"  tile {r}: k-tiles 0-2: {kt[0]:5.2f} {kt[1]:5.2f} {kt[2]:5.2f}  steady {steady:5.2f}/k-tile  main loop {main:6.2f}  pre-barrier {bar:5.2f}  "
              f"epilogue {ep:5.2f}  post-barrier {bar2:5.2f}  to next k-tile {nxt:5.2f} us")
