mkdir -p gpurun_out/r5c; rm -f gpurun_out/r5c/single.txt
export NKBHIP_LIB=$PWD/build/alt_stamps_nostag/libnkbhip.so
for cfg in "256 768 768 plain" "2048 768 768 plain" "8192 768 768 plain" "16384 768 768 plain" "256 768 768 add" "8192 768 768 add"; do
  python scripts/g8_stamps.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/r5c/single.txt
done
cat gpurun_out/r5c/single.txt
