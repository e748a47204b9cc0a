mkdir -p gpurun_out/r5f; rm -rf gpurun_out/r5f/* gpurun_out/pmc1_*
for r in 1 2; do
  for g in 8 0 4 16; do
    if [ $g = 8 ]; then unset NKBHIP_LIB; else export NKBHIP_LIB=$PWD/build/alt_gm$g/libnkbhip.so; fi
    TAG=gm$g python scripts/gemm8p_epi_bench.py 12 2>&1 | grep -v amdgpu > gpurun_out/r5f/epi_gm${g}_$r.txt
  done
done
unset NKBHIP_LIB
for g in 8 0 4 16; do echo "== group_m $g"; head -5 gpurun_out/r5f/epi_gm${g}_2.txt | cut -c1-62; tail -n 1 gpurun_out/r5f/epi_gm${g}_1.txt gpurun_out/r5f/epi_gm${g}_2.txt | grep chain; done
# SQ instruction mix of the forward GEMM on the two common shapes (default build)
bash scripts/pmc_one.sh g8_qkv gemm8p one_gemm.py 50432 768 2304 > gpurun_out/r5f/sq_qkv.txt 2>&1
bash scripts/pmc_one.sh g8_fc2 gemm8p one_gemm.py 50432 3072 768 > gpurun_out/r5f/sq_fc2.txt 2>&1
cat gpurun_out/r5f/sq_qkv.txt
# fabric-side traffic per launch under the four walks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 8 0 4 16; do
  if [ $g = 8 ]; then unset NKBHIP_LIB; else export NKBHIP_LIB=$R/build/alt_gm$g/libnkbhip.so; fi
  for shape in "50432 768 2304" "50432 768 3072"; do
    for C in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r5f/pmc_gm${g}_$C -o p -- python3 $R/scripts/one_gemm.py $shape > /dev/null 2>&1
      python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/r5f/pmc_gm${g}_$C/**/p_counter_collection.csv", recursive=True):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm8p" in r["Kernel_Name"]]
    if v: print("group_m $g shape $shape $C KiB/launch", sum(v[-3:])/len(v[-3:]))
PY
      rm -rf $R/gpurun_out/r5f/pmc_gm${g}_$C
    done
  done
done
