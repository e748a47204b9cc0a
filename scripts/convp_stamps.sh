#!/bin/bash
# Diagnostic build of libnkbhip with in-kernel cycle stamps in the row-balanced 3x3 core (NKB_CONVP_STAMPS), run on one shape.
# usage (on the GPU box): bash scripts/convp_stamps.sh <N> <H> <Cin> <Cout> <kind>
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/stampbuild && cd $R/nkb-classification_amd/csrc || exit 1
for f in *.hip; do o=/tmp/stampbuild/${f%.hip}.o; cp ../lib/obj/${f%.hip}.o $o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -Wno-unused-value -ffp-contract=off -fno-slp-vectorize -DNKB_CONVP_STAMPS -c convp.hip -o /tmp/stampbuild/convp.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/stampbuild/*.o -o /tmp/stampbuild/libnkbhip_stamps.so || exit 1
cd $R && NKBHIP_LIB=/tmp/stampbuild/libnkbhip_stamps.so python3 - "$@" <<'PY'
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "nkb-classification_amd"))
from nkb_classification import hip
N, H, ci, co, kind = [int(v) for v in sys.argv[1:6]]
dev, T, d = "cuda", torch.bfloat16, hip.BF16
x = torch.randn(N, H, H, ci, device=dev).to(T); w = (torch.randn(co, 3, 3, ci, device=dev) * 0.05).to(T)
y = torch.empty(N, H, H, co, device=dev, dtype=T)
tiles = hip.convp_tiles(d, kind, N=N, H=H, W=H, Cin=ci, ldx=ci, Cout=co, ldy=co, R=3, S=3, stride=1, pad=1)
st = torch.zeros(hip.bn_stats_floats(tiles, co), device=dev)
c = torch.randn(N, H, H, co, device=dev).to(T); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
for _ in range(3):
    if kind == 0: hip.convp_fwd(d, x, w, y, st, N=N, H=H, W=H, Cin=ci, ldx=ci, Cout=co, ldy=co)
    else: hip.convp_dgrad_bn(d, x, w, y, c, sc, sh, sh, st, N=N, H=H, W=H, Cin=ci, ldx=ci, Cout=co, ldy=co)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib = ctypes.CDLL(os.environ["NKBHIP_LIB"])
assert lib.nkb_convp_read_stamps(buf) == 0
names = ["between k-tiles", "vmcnt wait / barrier / X issue", "filter fragments", "filter DMA issue", "pixel fragments + MFMA", "tail"]
if ci == 64 and co == 64:      # convp64_kernel: per chunk
    names = ["between chunks", "vmcnt wait", "barrier", "rows + X issue", "fragments + MFMA (3 k-tiles)", "epilogue"]
kt = 36 if ci == 256 else (72 if ci == 512 else 18 * ((N * H * H // 256 + 255) // 256))
for wv in range(2):
    tot = sum(buf[wv * 8 + i] for i in range(6))
    print(f"wave {4 * wv}: total {tot} cycles (s_memtime ticks)")
    for i, n in enumerate(names):
        print(f"   {n:34s} {buf[wv * 8 + i]:9d}  {100.0 * buf[wv * 8 + i] / max(tot, 1):5.1f} %")
PY
