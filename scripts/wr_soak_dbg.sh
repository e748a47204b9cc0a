#!/bin/bash
# Determinism soak of the train step with diagnostic builds of wgradr (GPU box): variants = extra -D flags, e.g. "-DNKB_WR_SAFE_VM"
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/diagbuild && cd $R/nkb-classification_amd/csrc || exit 1
for f in *.hip; do cp ../lib/obj/${f%.hip}.o /tmp/diagbuild/${f%.hip}.o; done
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -Wno-unused-value -ffp-contract=off -fno-slp-vectorize $v -c wgradr.hip -o /tmp/diagbuild/wgradr.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/diagbuild/*.o -o /tmp/diagbuild/libnkbhip_diag.so || exit 1
  echo "== [$v]"
  (cd $R && NKBHIP_LIB=/tmp/diagbuild/libnkbhip_diag.so timeout -k 10 300 python scripts/soak_determinism.py --steps ${SOAK_STEPS:-6} --runs 4 2>&1 | tail -4)
done
