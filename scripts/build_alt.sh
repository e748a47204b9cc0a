#!/bin/bash
# An alternative build of libnkbhip.so for same-process-pool A/B runs (NKBHIP_LIB=build/alt_<name>/libnkbhip.so): the listed sources are
# recompiled with the extra flags, every other object is the default build's.  Usage: scripts/build_alt.sh <name> "<flags>" file.hip [...]
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/nkb-classification_amd/csrc; O=$R/build/alt_$NAME
mkdir -p $O
make -C $C -j8 > /dev/null
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f .hip); obj=$R/nkb-classification_amd/lib/obj/$b.o
  for s in "$@"; do if [ "$s" = "$b.hip" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$C -I$R/include -Wno-unused-result -Wno-unused-value -Wno-inline-asm -ffp-contract=off -fno-slp-vectorize $FLAGS -c $f -o $O/$b.o
    obj=$O/$b.o
  fi; done
  OBJS="$OBJS $obj"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o $O/libnkbhip.so
echo built $O/libnkbhip.so
