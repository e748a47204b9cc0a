mkdir -p gpurun_out/r5e; rm -f gpurun_out/r5e/*
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "bn_ or batchnorm or stats or wgradr_split" > gpurun_out/r5e/t1.txt 2>&1; tail -3 gpurun_out/r5e/t1.txt
python -m pytest tests/test_parity_bench_size_gpu.py -x -q -m gpu -k "resnet" > gpurun_out/r5e/t2.txt 2>&1; tail -3 gpurun_out/r5e/t2.txt
for r in 1 2 3; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused', d['ms_per_step'], d['host_work_ms_per_step'], d['final_loss'])"
  NKBHIP_LIB=$PWD/build/alt_bn2/libnkbhip.so python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two  ', d['ms_per_step'], d['host_work_ms_per_step'], d['final_loss'])"
done
