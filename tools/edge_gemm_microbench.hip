// Microbenchmark of the complete 8-wave edge GEMM (w8_common.h: edge_gemm_pq): barrier, ring staging from a weight buffer
// in L2 / Infinity Cache, on-the-fly input generation from LDS rows, MFMAs.  Prints cycles per trip against the MFMA floor.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaudi_amd/csrc tools/edge_gemm_microbench.hip -o egemm_mb && ./egemm_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "w8_common.h"
using namespace gaudi;

template <int HP>
__global__ __launch_bounds__(512) void k(const float* w, unsigned wbytes, int nmat, float* out, unsigned long long* cyc, int gemms,
                                         int nactive) {
  constexpr int T = HP / 16, LD = HP + 4, N = 11;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring_mem = smem;
  float* sP = ring_mem + 2 * T * 256;
  float* sQ = sP + N * LD;
  float* vec = sQ + N * LD;  // b2 | cr | cd
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
  for (int i = tid; i < 2 * N * LD + 3 * HP; i += 512) sP[i] = 0.01f * ((i * 7) % 13) - 0.05f;
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  w8::Ring<HP> ring;
  ring.base = ring_mem;
  ring.par = 0;
  ring.ktail = false;
  w8::ring_start<HP>(ring, wb, 0, wave, lane);
  const bool active = wave < nactive;
  const int i = (wave + c) % N, j = (wave * 3 + c) % N;
  f4 total = splat(0.f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * T * T * 256, nextW = ((m + 1) % nmat) * T * T * 256;
    f4 acc[T];
    w8::edge_gemm_pq<HP>(acc, ring, wb, W, nextW, vec, vec + HP, vec + 2 * HP, sP + i * LD + 4 * g, sQ + j * LD + 4 * g, 0.3f,
                         0.7f, active, wave, lane);
#pragma unroll
    for (int t = 0; t < T; ++t) total = total + acc[t];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = total[0] + total[1] + total[2] + total[3];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int HP>
void run(int nactive, int blocks, int nmat) {
  constexpr int T = HP / 16;
  float *out, *w;
  unsigned long long* cyc;
  const size_t wfloats = (size_t)nmat * T * T * 256;
  hipMalloc(&w, wfloats * 4);
  hipMemset(w, 0, wfloats * 4);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int gemms = 200;
  const size_t lds = (2 * T * 256 + 2 * 11 * (HP + 4) + 3 * HP) * 4;
  hipFuncSetAttribute((const void*)k<HP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((k<HP>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, cyc, gemms, nactive);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (double)h[wv]);
  printf("edge_gemm_pq HP=%d active=%d blocks=%d matrices=%d (%.1f MB): %.0f cycles per trip; MFMA floor %d\n", HP, nactive, blocks,
         nmat, wfloats * 4 / 1e6, mx / gemms / T, T * 4 * 32 * (nactive > 4 ? 2 : 1));
  hipFree(out);
  hipFree(cyc);
  hipFree(w);
}

int main(int argc, char** argv) {
  if (argc > 1) {  // quick mode
    run<192>(8, 256, 27);
    run<192>(4, 256, 27);
    return 0;
  }
  run<192>(8, 256, 27);
  run<192>(7, 256, 27);
  run<192>(8, 1, 27);
  run<192>(8, 256, 1);
  run<192>(4, 256, 27);
  run<208>(7, 256, 72);
  return 0;
}
