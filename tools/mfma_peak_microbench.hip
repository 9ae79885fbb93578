#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
  f4 acc[4] = {{0, 0, 0, 1}, {0, 0, 0, 2}, {0, 0, 0, 3}, {0, 0, 0, 4}};
  bf8 x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)a; y[i] = (__bf16)b; }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (KIND == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[r & 3], 0, 0, 0);
      if (KIND == 1) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r & 3], 0, 0, 0);
      if (KIND == 2) acc[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[r & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(int threads, double flop_per_inst, const char* name) {
  float* out;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double inst = 256.0 * (threads / 64) * iters * 16.0;
  printf("%s, %d waves per CU: %.3f ms, %.1f TFLOP/s, %.2f ns per instruction per SIMD\n", name, threads / 64, ms, inst * flop_per_inst / (ms * 1e-3) / 1e12,
         ms * 1e6 / (inst / 1024.0));
  (void)hipFree(out);
}
int main() {
  for (int t : {256, 512, 1024}) run<0>(t, 16384.0, "v_mfma_f32_16x16x32_bf16");
  for (int t : {256, 512, 1024}) run<1>(t, 2048.0, "v_mfma_f32_16x16x4_f32  ");
  for (int t : {256, 512, 1024}) run<2>(t, 512.0, "v_mfma_f32_4x4x1_16B_f32");
  return 0;
}
