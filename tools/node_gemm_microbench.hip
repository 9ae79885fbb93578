// Microbenchmark of the 8-wave node-level GEMM (w8_common.h: node_gemm): weights streamed from L2 / Infinity Cache, one
// matrix after the other with a barrier in between (as the layers do), N = 11 or 20 graph nodes in LDS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaudi_amd/csrc tools/node_gemm_microbench.hip -o ngemm_mb && ./ngemm_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "w8_common.h"
using namespace gaudi;

template <int HP>
__global__ __launch_bounds__(512) void k(const float* w, unsigned wbytes, int nmat, float* out, unsigned long long* cyc, int gemms,
                                         int N) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;
  float* sY = sX + 20 * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 20 * LD; i += 512) smem[i] = 0.01f * ((i * 7) % 13) - 0.05f;
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  w8::NodePF<HP> pf;
  w8::node_prefetch<HP>(pf, wb, 0, wave, lane);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * T * T * 256, nextW = ((m + 1) % nmat) * T * T * 256;
    w8::node_gemm<HP, EPI_SILU, true>(wb, W, (m & 1) ? sY : sX, -1, nullptr, nullptr, (m & 1) ? sX : sY, nullptr, nullptr, N, wave,
                                      lane, &pf, nextW);
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = sX[tid % (20 * LD)];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int HP>
void run(int N, int blocks, int nmat) {
  constexpr int T = HP / 16;
  float *out, *w;
  unsigned long long* cyc;
  const size_t wfloats = (size_t)nmat * T * T * 256;
  hipMalloc(&w, wfloats * 4);
  hipMemset(w, 0, wfloats * 4);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int gemms = 600;
  const size_t lds = 2 * 20 * (HP + 4) * 4;
  hipFuncSetAttribute((const void*)k<HP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((k<HP>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, cyc, gemms, N);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (double)h[wv]);
  const int crit = ((T + 7) / 8 + (T > 4 ? (T - 4 + 7) / 8 : 0));  // tiles on SIMD 0 (waves 0 and 4)
  printf("node_gemm HP=%d N=%d blocks=%d matrices=%d (%.1f MB): %.0f cycles per matrix; MFMA floor %d (SIMD 0: %d tiles), %d KiB of weights\n",
         HP, N, blocks, nmat, wfloats * 4 / 1e6, mx / gemms, crit * T * 4 * 32 * ((N + 15) / 16), crit, T * T);
  hipFree(out);
  hipFree(cyc);
  hipFree(w);
}

int main() {
  run<192>(11, 256, 63);
  run<208>(11, 256, 120);
  run<208>(11, 1, 120);
  run<208>(11, 256, 1);
  run<208>(20, 256, 120);
  run<256>(11, 256, 60);
  return 0;
}
