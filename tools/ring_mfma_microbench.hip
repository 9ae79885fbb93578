// Microbenchmark of the 8-wave edge-GEMM inner block (w8_common.h: ring_mfma): T tiles of A fragments read from an LDS
// slot, 4 MFMAs per tile, two waves per SIMD.  Prints cycles per trip (T*4 MFMAs per wave) for a few variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaudi_amd/csrc tools/ring_mfma_microbench.hip -o ring_mb && ./ring_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "w8_common.h"
using namespace gaudi;

template <int HP, int VAR>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int trips, int nwaves_active) {
  constexpr int T = HP / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * T * 256; i += 512) smem[i] = 0.001f * (i & 255);
  __syncthreads();
  f4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = splat(0.f);
  f4 bin = {1.f + lane, 2.f, 3.f, 4.f};
  const bool active = wave < nwaves_active;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int par = 0;
#pragma unroll 1
  for (int it = 0; it < trips; ++it) {
    if (VAR == 1) __syncthreads();
    if (active) {
      const float* slot = smem + par * T * 256 + lane * 4;
      if (VAR == 2) {  // A fragments from registers: no LDS traffic at all
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += 2)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[t0] = mfma1(bin[q], bin[q], acc[t0]);
            if (t0 + 1 < T) acc[t0 + 1] = mfma1(bin[(q + 1) & 3], bin[q], acc[t0 + 1]);
          }
      } else {
        w8::ring_mfma<HP, 4>(acc, slot, bin, true, [] {});
      }
    }
    par ^= 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f4 s = splat(0.f);
#pragma unroll
  for (int t = 0; t < T; ++t) s = s + acc[t];
  out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int HP, int VAR>
void run(const char* name, int nactive, int blocks) {
  constexpr int T = HP / 16;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int trips = 2000;
  const size_t lds = 2 * T * 256 * 4;
  hipFuncSetAttribute((const void*)k<HP, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<HP, VAR>), dim3(blocks), dim3(512), lds, 0, out, cyc, trips, nactive);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int w = 0; w < 8; ++w) mx = std::max(mx, (double)h[w]);
  printf("%-44s HP=%d active waves=%d blocks=%d: %.0f cycles per trip (slowest wave of block 0); ideal %d (%d MFMA x 32 x %d waves/SIMD)\n",
         name, HP, nactive, blocks, mx / trips, T * 4 * 32 * (nactive > 4 ? 2 : 1), T * 4, nactive > 4 ? 2 : 1);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<192, 0>("ring_mfma, no barrier", 8, 256);
  run<192, 0>("ring_mfma, no barrier", 7, 256);
  run<192, 0>("ring_mfma, no barrier", 4, 256);
  run<192, 0>("ring_mfma, no barrier", 8, 1);
  run<192, 1>("ring_mfma + barrier per trip", 8, 256);
  run<192, 2>("MFMAs on registers only", 8, 256);
  run<192, 2>("MFMAs on registers only", 4, 256);
  run<208, 0>("ring_mfma, no barrier", 8, 256);
  run<208, 2>("MFMAs on registers only", 8, 256);
  return 0;
}
