set -x
mkdir -p gpurun_out/r5b
python -m pytest tests/test_convp_gpu.py tests/test_conv1p_stemp_gpu.py -x -q -m gpu > gpurun_out/r5b/t1.txt 2>&1; tail -3 gpurun_out/r5b/t1.txt
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgradr or gemm8p or linear" > gpurun_out/r5b/t2.txt 2>&1; tail -3 gpurun_out/r5b/t2.txt
for r in 1 2; do
  TAG=new python scripts/gemm8p_epi_bench.py 12 > gpurun_out/r5b/epi_new_$r.txt 2>&1
  TAG=old NKBHIP_LIB=$PWD/build/alt_preold/libnkbhip.so python scripts/gemm8p_epi_bench.py 12 > gpurun_out/r5b/epi_old_$r.txt 2>&1
done
tail -10 gpurun_out/r5b/epi_new_2.txt; tail -10 gpurun_out/r5b/epi_old_2.txt
