mkdir -p gpurun_out/r06r
B="python3 bench.py --batch 1024 --diffusion-steps 100 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity-gate"
for pr in 0 1; do
  GAUDI_LIB=$PWD/gaudi_amd/libgaudi_hip_stamps.so GAUDI_PRINT_STAMPS=1 GAUDI_PAIRS=$pr timeout 600 $B > gpurun_out/r06r/stamps_pairs$pr.json 2> gpurun_out/r06r/stamps_pairs$pr.err
  grep "stamps" gpurun_out/r06r/stamps_pairs$pr.err | tail -2 > gpurun_out/r06r/stamps_pairs$pr.txt
done
GAUDI_LIB=$PWD/gaudi_amd/libgaudi_hip_stamps.so GAUDI_PRINT_STAMPS=1 timeout 600 python3 bench.py --diffusion-steps 100 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity-gate > gpurun_out/r06r/stamps_c3.json 2> gpurun_out/r06r/stamps_c3.err
grep "stamps" gpurun_out/r06r/stamps_c3.err | tail -2 > gpurun_out/r06r/stamps_c3.txt
cat gpurun_out/r06r/stamps_*.txt
