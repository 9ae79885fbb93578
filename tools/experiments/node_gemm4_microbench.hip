// Microbenchmark + numerics probe of the node-level GEMMs: V = 0 the fp32-instruction form on 8 waves (w8_common.h: node_gemm),
// V = 1 the split-operand form on 4 waves (tools/experiments/w8_nodes_split.h: node_gemm4; measured and rejected).  Weights streamed from L2 /
// Infinity Cache, one matrix after the other with a barrier in between (as the layers do), N = 11 or 20 graph nodes in LDS.
// `tail` = the 4-valid-row output tile of H % 16 == 4 widths computed with v_mfma_f32_4x4x1_16B_f32 (w8_common.h: tail_lane).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I gaudi_amd/csrc -I tools/experiments tools/node_gemm4_microbench.hip -o gaudi_amd/ngemm4_mb && ./ngemm_mb
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "w8_nodes_split.h"  // tools/experiments
using namespace gaudi;

template <int HP, int V>
struct Sel;
template <int HP>
struct Sel<HP, 0> {
  using PF = w8::NodePF<HP>;
  static __device__ __forceinline__ void prefetch(PF& pf, const WBuf& wb, int W, int wave, int lane, bool tw) { w8::node_prefetch<HP>(pf, wb, W, wave, lane, tw); }
  template <int EPI>
  static __device__ __forceinline__ void gemm(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb, const float* sB, float* sY, int N,
                                              int wave, int lane, bool tw, PF* pf, int nextW) {
    w8::node_gemm<HP, EPI, true>(wb, Wa, sXa, Wb, sXb, sB, sY, nullptr, nullptr, N, wave, lane, tw, pf, nextW);
  }
};
template <int HP>
struct Sel<HP, 1> {
  using PF = w8::NodePF4<HP>;
  static __device__ __forceinline__ void prefetch(PF& pf, const WBuf& wb, int W, int wave, int lane, bool tw) { w8::node_prefetch4<HP>(pf, wb, W, wave, lane, tw); }
  template <int EPI>
  static __device__ __forceinline__ void gemm(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb, const float* sB, float* sY, int N,
                                              int wave, int lane, bool tw, PF* pf, int nextW) {
    w8::node_gemm4<HP, EPI, true>(wb, Wa, sXa, Wb, sXb, sB, sY, nullptr, nullptr, N, wave, lane, tw, pf, nextW);
  }
};

// GX: the activations (node buffers) live in a per-workgroup GLOBAL scratch instead of LDS (the V8G kernels of round 4)
template <int HP, int V, bool GX = false>
__global__ __launch_bounds__(512) void k(const float* w, unsigned wbytes, int nmat, float* out, unsigned long long* cyc, int gemms,
                                         int N, int tail, float* gscratch = nullptr) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = GX ? gscratch + (size_t)blockIdx.x * 2 * 48 * LD : smem;
  float* sY = sX + 48 * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 48 * LD; i += 512) sX[i] = 0.01f * ((i * 7) % 13) - 0.05f;
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  typename Sel<HP, V>::PF pf;
  Sel<HP, V>::prefetch(pf, wb, 0, wave, lane, tail != 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * T * T * 256, nextW = ((m + 1) % nmat) * T * T * 256;
    Sel<HP, V>::template gemm<EPI_SILU>(wb, W, (m & 1) ? sY : sX, -1, nullptr, nullptr, (m & 1) ? sX : sY, N, wave, lane, tail != 0, &pf, nextW);
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = sX[tid % (48 * LD)];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// numerics: Y[n][o] = bias[o] + sum_k Wa[o][k] Xa[n][k] + sum_k Wb[o][k] Xb[n][k]   (one workgroup, EPI_NONE, two sources)
template <int HP, int V>
__global__ __launch_bounds__(512) void k_num(const float* w, unsigned wbytes, const float* x, const float* bias, float* y, int N,
                                             int tail, int two) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sXa = smem;
  float* sXb = sXa + 32 * LD;
  float* sY = sXb + 32 * LD;
  float* sB = sY + 32 * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 32 * LD; i += 512) {
    const int n = i / LD, f = i % LD;
    sXa[i] = (n < N && f < HP) ? x[n * HP + f] : 0.f;
    sXb[i] = (n < N && f < HP) ? x[(32 + n) * HP + f] : 0.f;
    sY[i] = __builtin_nanf("");  // every feature of every live node must be written
  }
  for (int i = tid; i < HP; i += 512) sB[i] = bias[i];
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  typename Sel<HP, V>::PF pf;
  Sel<HP, V>::prefetch(pf, wb, 0, wave, lane, tail != 0);
  Sel<HP, V>::template gemm<EPI_NONE>(wb, 0, sXa, two ? T * T * 256 : -1, sXb, sB, sY, N, wave, lane, tail != 0, &pf, -1);
  __syncthreads();
  for (int i = tid; i < N * HP; i += 512) y[i] = sY[(i / HP) * LD + i % HP];
}

template <int HP, int V, bool GX = false>
void run(int N, int blocks, int nmat, int tail) {
  constexpr int T = HP / 16;
  float *out, *w;
  unsigned long long* cyc;
  const size_t wfloats = (size_t)nmat * T * T * 256;
  hipMalloc(&w, wfloats * 4);
  hipMemset(w, 0, wfloats * 4);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int gemms = 600;
  const size_t lds = GX ? 64 : 2 * 48 * (HP + 4) * 4;
  float* gs = nullptr;
  if (GX) hipMalloc(&gs, (size_t)blocks * 2 * 48 * (HP + 4) * 4);
  hipFuncSetAttribute((const void*)k<HP, V, GX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<HP, V, GX>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, cyc, gemms, N, tail, gs);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (double)h[wv]);
  const int crit = ((T + 7) / 8 + (T > 4 ? (T - 4 + 7) / 8 : 0));  // tiles on SIMD 0 (waves 0 and 4)
  printf("%s HP=%d N=%d blocks=%d matrices=%d (%.1f MB) tail44=%d: %.0f cycles, %.3f us per matrix; fp32 MFMA floor of the 8-wave form %d (SIMD 0: %d tiles)\n",
         V ? "node_gemm4 (4 waves, split operands)" : GX ? "node_gemm  (8 waves, fp32 MFMA, activations in GLOBAL memory)" : "node_gemm  (8 waves, fp32 MFMA)     ", HP, N, blocks, nmat, wfloats * 4 / 1e6, tail, mx / gemms, ms * 1e3 / gemms,
         crit * T * 4 * 32 * ((N + 15) / 16), crit);
  hipFree(out);
  hipFree(cyc);
  hipFree(w);
}

// lane-linear fp32 tiles [k/16][o/16], float4 index L = (row L & 15, k-quad L >> 4)
template <int HP>
static void pack_f32(float* dst, const std::vector<float>& W, int H) {
  constexpr int T = HP / 16;
  for (int kc = 0; kc < T; ++kc)
    for (int t = 0; t < T; ++t)
      for (int L = 0; L < 64; ++L)
        for (int q = 0; q < 4; ++q) {
          const int o = 16 * t + (L & 15), kk = 16 * kc + 4 * (L >> 4) + q;
          dst[((size_t)(kc * T + t) * 64 + L) * 4 + q] = (o < H && kk < H) ? W[(size_t)o * H + kk] : 0.f;
        }
}

template <int HP, int V>
void run_num(int H, int N, int tail, int two) {
  constexpr int T = HP / 16;
  std::mt19937 rng(11 + N);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> Wa((size_t)H * H), Wb((size_t)H * H), x((size_t)64 * HP, 0.f), bias(HP, 0.f);
  for (auto& v : Wa) v = nd(rng) / std::sqrt((float)H);
  for (auto& v : Wb) v = nd(rng) / std::sqrt((float)H);
  for (int n = 0; n < 64; ++n)
    for (int f = 0; f < H; ++f) x[(size_t)n * HP + f] = nd(rng);
  for (int f = 0; f < H; ++f) bias[f] = nd(rng);
  std::vector<float> pk((size_t)2 * T * T * 256, 0.f);
  pack_f32<HP>(pk.data(), Wa, H);
  pack_f32<HP>(pk.data() + (size_t)T * T * 256, Wb, H);
  float *dw, *dx, *db, *dy;
  hipMalloc(&dw, pk.size() * 4);
  hipMalloc(&dx, x.size() * 4);
  hipMalloc(&db, HP * 4);
  hipMalloc(&dy, (size_t)N * HP * 4);
  hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), HP * 4, hipMemcpyHostToDevice);
  const size_t lds = (3 * 32 * (HP + 4) + HP) * 4;
  hipFuncSetAttribute((const void*)k_num<HP, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_num<HP, V>), dim3(1), dim3(512), lds, 0, dw, (unsigned)(pk.size() * 4), dx, db, dy, N, tail, two);
  std::vector<float> y((size_t)N * HP);
  hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
  double emax = 0, rmax = 0, padmax = 0;
  int nan = 0;
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < HP; ++o) {
      const float got = y[(size_t)n * HP + o];
      if (got != got) { ++nan; continue; }
      if (o >= H) { padmax = std::max(padmax, (double)std::fabs(got)); continue; }
      double s = bias[o];
      for (int kk = 0; kk < H; ++kk) {
        s += (double)Wa[(size_t)o * H + kk] * x[(size_t)n * HP + kk];
        if (two) s += (double)Wb[(size_t)o * H + kk] * x[(size_t)(32 + n) * HP + kk];
      }
      emax = std::max(emax, std::fabs(s - got));
      rmax = std::max(rmax, std::fabs(s));
    }
  printf("numerics %s H=%d HP=%d N=%d sources=%d tail44=%d: max|err| vs float64 %.3e of max|ref| %.3e (%.2e rel); padding max %.1e; NaN %d  %s\n",
         V ? "node_gemm4" : "node_gemm ", H, HP, N, 1 + two, tail, emax, rmax, emax / rmax, padmax, nan, (emax / rmax < 2e-6 && padmax == 0 && nan == 0) ? "OK" : "FAIL");
  hipFree(dw); hipFree(dx); hipFree(db); hipFree(dy);
}

template <int V>
void all() {
  for (int tail = 0; tail < 2; ++tail) {
    run_num<208, V>(196, 11, tail, 0);
    run_num<208, V>(196, 11, tail, 1);
    run_num<208, V>(196, 22, tail, 1);
    run_num<48, V>(36, 7, tail, 1);
    run_num<48, V>(36, 18, tail, 0);
  }
  run_num<208, V>(208, 11, 0, 1);
  run_num<192, V>(192, 11, 0, 0);
  run_num<192, V>(192, 22, 0, 1);
  run_num<32, V>(32, 5, 0, 1);
  run_num<64, V>(64, 9, 0, 0);
  run_num<128, V>(128, 20, 0, 1);
  run_num<256, V>(256, 11, 0, 1);
  run<192, V>(11, 256, 63, 0);
  run<192, V>(22, 256, 63, 0);
  run<208, V>(11, 256, 120, 1);
  run<208, V>(11, 1, 120, 1);
  run<208, V>(20, 256, 120, 1);
  run<208, V>(22, 256, 120, 1);
  run<208, V>(11, 256, 1, 1);  // the weights resident in L2 (one matrix)
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'g') {  // activations in LDS vs in global memory (V8G), fp32 node GEMM, N = 22 / 32 / 40
    for (int N : {22, 32, 40}) {
      run<208, 0, false>(N, 256, 120, 1);
      run<208, 0, true>(N, 256, 120, 1);
      run<192, 0, false>(N, 256, 63, 0);
      run<192, 0, true>(N, 256, 63, 0);
    }
    return 0;
  }
  if (argc > 1) {  // one timing configuration (for counter passes): variant N
    if (atoi(argv[1])) run<208, 1>(argc > 2 ? atoi(argv[2]) : 11, 256, 120, 1);
    else run<208, 0>(argc > 2 ? atoi(argv[2]) : 11, 256, 120, 1);
    return 0;
  }
  all<0>();
  all<1>();
  return 0;
}
