mkdir -p gpurun_out/r5g; rm -rf gpurun_out/r5g/*
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm8p or linear or fp8" > gpurun_out/r5g/t.txt 2>&1; tail -3 gpurun_out/r5g/t.txt
for r in 1 2 3; do
  TAG=sdma python scripts/gemm8p_epi_bench.py 12 2>&1 | grep -v amdgpu > gpurun_out/r5g/epi_sdma_$r.txt
  TAG=old NKBHIP_LIB=$PWD/build/alt_nosdma/libnkbhip.so python scripts/gemm8p_epi_bench.py 12 2>&1 | grep -v amdgpu > gpurun_out/r5g/epi_old_$r.txt
done
paste gpurun_out/r5g/epi_sdma_2.txt gpurun_out/r5g/epi_old_2.txt | cut -c1-42,98-108
for f in gpurun_out/r5g/epi_*; do echo $f $(tail -n 1 $f); done
python scripts/gemm8p_bench.py 2>&1 | grep -v amdgpu | cut -c1-75 > gpurun_out/r5g/g8bench.txt; cat gpurun_out/r5g/g8bench.txt
