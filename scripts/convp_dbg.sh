#!/bin/bash
# Ablation timing of the row-balanced 3x3 core in a DIAGNOSTIC build (-DNKB_CONVP_DIAG: the NKB_CONVP_DBG knobs exist only there):
# bit 0 no activation DMA, 1 no filter DMA, 2 no MFMA (garbage results), 6 (64) no rotated wave group, 7 (128) no s_setprio.
# usage (GPU box): [CHECK_MODE=c64] bash scripts/convp_dbg.sh [values ...]   (bit 3 (8): no forward stores)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/diagbuild && cd $R/nkb-classification_amd/csrc || exit 1
for f in *.hip; do cp ../lib/obj/${f%.hip}.o /tmp/diagbuild/${f%.hip}.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -Wno-unused-value -ffp-contract=off -fno-slp-vectorize -DNKB_CONVP_DIAG -c convp.hip -o /tmp/diagbuild/convp.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/diagbuild/*.o -o /tmp/diagbuild/libnkbhip_diag.so || exit 1
cd $R
for d in ${@:-0 1 2 3 4 7}; do
  echo "== NKB_CONVP_DBG=$d"
  NKBHIP_LIB=/tmp/diagbuild/libnkbhip_diag.so NKB_CONVP_DBG=$d timeout -k 10 200 python scripts/convp_check.py $CHECK_MODE 2>&1 | grep "N=256" | sed -E 's/^(kind [01] N=256 +[0-9x]+ +[0-9]+->[0-9]+).*(old +[0-9.]+ us +new +[0-9.]+ us).*/\1  \2/'
done
