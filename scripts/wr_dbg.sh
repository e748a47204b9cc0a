#!/bin/bash
# Ablation timing of wgradr in DIAGNOSTIC builds (compile-time: a run-time knob puts a branch around every MFMA):
# variants: base, nodma (-DNKB_WR_NO_DMA), nomfma (-DNKB_WR_NO_MFMA), none (both).  usage (GPU box): bash scripts/wr_dbg.sh [variants ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/diagbuild && cd $R/nkb-classification_amd/csrc || exit 1
for f in *.hip; do cp ../lib/obj/${f%.hip}.o /tmp/diagbuild/${f%.hip}.o; done
for v in ${@:-base nodma nomfma none}; do
  case $v in base) X="";; nodma) X="-DNKB_WR_NO_DMA";; nomfma) X="-DNKB_WR_NO_MFMA";; none) X="-DNKB_WR_NO_DMA -DNKB_WR_NO_MFMA";; *) X="$v";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -Wno-unused-value -ffp-contract=off -fno-slp-vectorize $X -c wgradr.hip -o /tmp/diagbuild/wgradr.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/diagbuild/*.o -o /tmp/diagbuild/libnkbhip_diag.so || exit 1
  echo "== $v"
  (cd $R && NKBHIP_LIB=/tmp/diagbuild/libnkbhip_diag.so timeout -k 10 200 python scripts/wr_check.py time $WR_SHAPES 2>&1 | grep -E "${WR_GREP:-l3 conv3|l4 conv1 |l2 conv1 }")
done
