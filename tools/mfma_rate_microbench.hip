// Issue rate of the fp32 matrix instructions used by the node GEMMs: cycles per instruction with NA independent
// accumulators, one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate_microbench.hip -o gaudi_amd/mfma_rate_mb && gaudi_amd/mfma_rate_mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND, int NA>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
  f4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = (f4){0.f, 0.f, 0.f, (float)i};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NA>
void run(int threads) {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 512 * 4);
  (void)hipMalloc(&cyc, 8 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<KIND, NA>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // s_memtime ticks at 100 MHz: convert with the shader clock measured by a known-latency loop is overkill here; report
  // ticks and the ratio between kinds
  printf("%s accumulators=%d waves/SIMD=%d: %.3f memtime ticks per instruction per wave\n", KIND ? "4x4x1_16B " : "16x16x4   ",
         NA, threads / 256, (double)h[0] / (iters * 8.0 * NA));
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0, 1>(256); run<0, 2>(256); run<0, 4>(256); run<0, 4>(512);
  run<1, 1>(256); run<1, 2>(256); run<1, 3>(256); run<1, 4>(256); run<1, 6>(256); run<1, 8>(256);
  run<1, 3>(512); run<1, 6>(512); run<1, 8>(512);
  return 0;
}
