#!/bin/bash
# GPU box: same-box A/B of the working tree's library against one with ONE source file taken from git HEAD's version stored at
# build/head_<file> (the box has no .git: save it first with `git show HEAD:nkb-classification_amd/csrc/<file> > build/head_<file>`).
# usage: ab_head_file.sh <file.hip> <bench args...>
F=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/diagbuild/src && cd $R/nkb-classification_amd/csrc || exit 1
for f in *.hip; do cp ../lib/obj/${f%.hip}.o /tmp/diagbuild/${f%.hip}.o; done
cp $R/build/head_$F /tmp/diagbuild/src/$F || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -Wno-unused-value -ffp-contract=off -fno-slp-vectorize -c /tmp/diagbuild/src/$F -o /tmp/diagbuild/${F%.hip}.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/diagbuild/*.o -o /tmp/diagbuild/libnkbhip_diag.so || exit 1
cd $R
for i in 1 2 3; do
  for lib in /tmp/diagbuild/libnkbhip_diag.so ""; do
    NKBHIP_LIB=${lib:-$R/nkb-classification_amd/lib/libnkbhip.so} python bench.py "$@" --steps 20 --warmup 6 --no-cpu-baseline --no-host-work --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:+HEAD}${lib:-tree}', d['ms_per_step'], d['value'])"
  done
done
