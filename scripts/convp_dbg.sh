#!/bin/bash
# timing experiments on the row-balanced 3x3 core: NKB_CONVP_DBG bit 0 no activation DMA, bit 1 no filter DMA, bit 2 no MFMA (results are garbage)
for d in ${@:-0 1 2 3 4 7}; do  # needs a -DNKB_CONVP_STAMPS build of convp.hip (scripts/convp_stamps.sh builds one)
  echo "== NKB_CONVP_DBG=$d"
  NKB_CONVP_DBG=$d timeout -k 10 200 python scripts/convp_check.py 2>&1 | grep "N=256" | sed -E 's/^(kind [01] N=256 +[0-9x]+ +[0-9]+->[0-9]+).*(old +[0-9.]+ us +new +[0-9.]+ us).*/\1  \2/'
done
