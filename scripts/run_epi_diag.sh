# A/B of gemm8p epilogue builds on one box: scripts/run_epi_diag.sh <reps> <name> [<name> ...]   (name: default | an alt build of scripts/build_alt.sh)
set -o pipefail
O=gpurun_out/epi_ab; mkdir -p $O; R=$1; shift
for r in $(seq 1 $R); do
 for v in "$@"; do
  if [ $v = default ]; then unset NKBHIP_LIB; else export NKBHIP_LIB=$PWD/build/alt_$v/libnkbhip.so; fi
  TAG=$v timeout -k 10 120 python scripts/gemm8p_epi_bench.py 12 > $O/${v}_$r.txt 2>&1 || exit 1
 done
done
grep -h "chain" $O/*_[0-9].txt
