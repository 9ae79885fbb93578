// Microbenchmark + numerics probe of the node-level GEMMs: V = 0 the fp32-instruction form (w8_common.h: node_gemm), V = 3 the
// fp16-pair form (w8_nodes_f16.h: node_gemm_h -- hi + 2^-11 lo pieces of both operands, three products on
// v_mfma_f32_16x16x32_f16, the same bytes per weight as fp32).  Weights streamed from L2 / Infinity Cache, one matrix after the
// other with a barrier in between (as the layers do).  Numerics against float64, including activations far outside fp16's range.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I gaudi_amd/csrc tools/node_gemm_h_microbench.hip -o gaudi_amd/ngemmh_mb
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "w8_nodes_f16.h"
using namespace gaudi;
#ifndef GAUDI_MB_AHEAD
#define GAUDI_MB_AHEAD w8::kAheadAll  // chunks that travel ahead of a call; w8::kAheadOne: what most call sites of the kernels use
#endif
#ifndef GAUDI_MB_FL
#define GAUDI_MB_FL false  // true: the FL form (lane addresses recomputed per call: the MR / GN / FR kernels)
#endif
#ifndef GAUDI_MB_MAXNT
#define GAUDI_MB_MAXNT 3  // column tiles per pass the fp16 form is instantiated for (2: the resident kernels; N = 40 runs are skipped)
#endif

template <int HP, int V>
struct Sel;
template <int HP>
struct Sel<HP, 0> {
  using PF = w8::NodePF<HP>;
  static __device__ __forceinline__ void prefetch(PF& pf, const WBuf& wb, int W, int wave, int lane, bool tw, int) { w8::node_prefetch<HP>(pf, wb, W, wave, lane, tw); }
  template <int EPI>
  static __device__ __forceinline__ void gemm(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb, const float* sB, float* sY, int N,
                                              int wave, int lane, bool tw, PF* pf, int nextW, float*, float, bool, w8::NodeStampH* = nullptr) {
    w8::node_gemm<HP, EPI, true>(wb, Wa, sXa, Wb, sXb, sB, sY, nullptr, nullptr, N, wave, lane, tw, pf, nextW);
  }
};
template <int HP>
struct Sel<HP, 3> {
  using PF = w8::NodePFH<HP>;
  static __device__ __forceinline__ void prefetch(PF& pf, const WBuf& wb, int W, int wave, int lane, bool tw, int N) { w8::node_prefetch_h<HP, GAUDI_MB_AHEAD>(pf, wb, W, wave, lane); }
  template <int EPI>
  static __device__ __forceinline__ void gemm(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb, const float* sB, float* sY, int N,
                                              int wave, int lane, bool tw, PF* pf, int nextW, float* split, float winv, bool seq,
                                              w8::NodeStampH* ns = nullptr) {
    const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
    w8::NodeCtxH cx{winv, split + 96, seq ? split + 96 : split + 96 + w8::nh_split_floats(HP, nct), tw, split};
    if (Wb >= 0)
      w8::node_gemm_h<HP, EPI, true, GAUDI_MB_MAXNT, GAUDI_MB_AHEAD, GAUDI_MB_AHEAD, GAUDI_MB_FL>(wb, Wa, sXa, true, Wb, sXb, sB, sY, nullptr, nullptr, N, wave, lane, cx, *pf, nextW,
                                                                                    nullptr, nullptr, ns);
    else
      w8::node_gemm_h<HP, EPI, false, GAUDI_MB_MAXNT, GAUDI_MB_AHEAD, GAUDI_MB_AHEAD, GAUDI_MB_FL>(wb, Wa, sXa, true, -1, nullptr, sB, sY, nullptr, nullptr, N, wave, lane, cx, *pf, nextW,
                                                                                     nullptr, nullptr, ns);
  }
};

template <int HP, int V>
__global__ __launch_bounds__(512) void k(const float* w, unsigned wbytes, int nmat, float* out, unsigned long long* cyc, int gemms,
                                         int N, int tail, int two) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int R = (N + 15) & ~15;
  float* sX = smem;
  float* sY = sX + R * LD;
  float* sSplit = smem + 2 * R * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * R * LD; i += 512) sX[i] = 0.01f * ((i * 7) % 13) - 0.05f;
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  const int MS = (V == 3 ? 1 : 1) * T * T * 256;  // matrix stride in fp32-offset units (the f16 images sit at 2 W)
  typename Sel<HP, V>::PF pf;
  Sel<HP, V>::prefetch(pf, wb, 0, wave, lane, tail != 0, N);
  w8::NodeStampH st;
  for (int i = 0; i < 8; ++i) st.sum[i] = 0;
  st.start();
  w8::NodeStampH* ns = GAUDI_NODE_STAMPS ? &st : nullptr;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * MS, nextW = ((m + 1) % nmat) * MS;
    if (two) {
      const int m2 = (m + nmat / 2) % nmat;
      Sel<HP, V>::template gemm<EPI_SILU>(wb, W, (m & 1) ? sY : sX, m2 * MS, (m & 1) ? sY : sX, nullptr, (m & 1) ? sX : sY, N, wave, lane, tail != 0, &pf,
                                          nextW, sSplit, 1.0f, two == 2, ns);
    } else {
      Sel<HP, V>::template gemm<EPI_SILU>(wb, W, (m & 1) ? sY : sX, -1, nullptr, nullptr, (m & 1) ? sX : sY, N, wave, lane, tail != 0, &pf, nextW, sSplit,
                                          1.0f, false, ns);
    }
    __syncthreads();
    if (GAUDI_NODE_STAMPS) st.mark(7);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = sX[tid % (R * LD)];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (GAUDI_NODE_STAMPS && lane == 0 && blockIdx.x == 0)
    for (int i = 0; i < 8; ++i) cyc[gridDim.x * 8 + wave * 8 + i] = st.sum[i];
}

// numerics: Y[n][o] = bias[o] + sum_k Wa[o][k] Xa[n][k] + sum_k Wb[o][k] Xb[n][k]   (one workgroup, EPI_NONE)
template <int HP, int V>
__global__ __launch_bounds__(512) void k_num(const float* w, unsigned wbytes, const float* x, const float* bias, float* y, int N,
                                             int tail, int two, float winv) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sXa = smem;
  float* sXb = sXa + N * LD;
  float* sY = sXb + N * LD;
  float* sB = sY + N * LD;
  float* sSplit = sB + ((HP + 63) / 64) * 64;
  const int nct_ = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < N * LD; i += 512) {
    const int n = i / LD, f = i % LD;
    sXa[i] = (n < N && f < HP) ? x[n * HP + f] : 0.f;
    sXb[i] = (n < N && f < HP) ? x[(48 + n) * HP + f] : 0.f;
    sY[i] = __builtin_nanf("");  // every feature of every live node must be written
  }
  for (int i = tid; i < HP; i += 512) sB[i] = bias[i];
  for (int i = tid; i < (two == 1 ? 2 : 1) * w8::nh_split_floats(HP, nct_) + 96; i += 512) sSplit[i] = __builtin_nanf("");  // stale ring contents
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  typename Sel<HP, V>::PF pf;
  Sel<HP, V>::prefetch(pf, wb, 0, wave, lane, tail != 0, N);
  Sel<HP, V>::template gemm<EPI_NONE>(wb, 0, sXa, two ? T * T * 256 : -1, sXb, sB, sY, N, wave, lane, tail != 0, &pf, -1, sSplit, winv, two == 2);
  __syncthreads();
  for (int i = tid; i < N * HP; i += 512) y[i] = sY[(i / HP) * LD + i % HP];
}

static uint16_t f16_rne(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
static float f16_to_f(uint16_t b) {
  _Float16 h;
  std::memcpy(&h, &b, 2);
  return (float)h;
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
// the fp16-pair image (w8_nodes_f16.h; gaudi_hip.hip: pack_matrix_f16): units [chunk][tile][piece][lane L = (row, 8 inputs)]
template <int HP>
static void pack_f16(float* dst, const std::vector<float>& W, int H, bool, float scale) {
  constexpr int T = HP / 16;
  uint16_t* d = (uint16_t*)dst;
  for (int o = 0; o < H; ++o)
    for (int k = 0; k < H; ++k) {
      const float v = W[(size_t)o * H + k];
      const int t = o / 16, i = o % 16;
      if (w8::nh_odd(HP) && k >= 16 * (T - 1)) {
        const int kk = k - 16 * (T - 1);
        dst[(size_t)w8::nh_tail_off(HP) + t * 256 + (kk / 4) * 64 + (kk % 4) * 16 + i] = v;
        continue;
      }
      const int m = k / 32, g = (k % 32) / 8, e = k % 8, L = g * 16 + i;
      const float vs = v * scale;
      const uint16_t hi = f16_rne(vs);
      const uint16_t lo = f16_rne((vs - f16_to_f(hi)) * 2048.f);
      d[(((size_t)(m * T + t) * 2 + 0) * 64 + L) * 8 + e] = hi;
      d[(((size_t)(m * T + t) * 2 + 1) * 64 + L) * 8 + e] = lo;
    }
}

template <int HP, int V>
void run(int N, int blocks, int nmat, int tail, int two = 0) {
  constexpr int T = HP / 16;
  float *out, *w;
  unsigned long long* cyc;
  const size_t wfloats = (size_t)nmat * T * T * 256 * (V == 3 ? 2 : 1);
  hipMalloc(&w, wfloats * 4);
  hipMemset(w, 0, wfloats * 4);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, (blocks * 8 + 64) * 8);
  const int gemms = 600;
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  const size_t lds = (2 * ((N + 15) & ~15) * (HP + 4) + (two == 1 ? 2 : 1) * w8::nh_split_floats(HP, nct) + 96) * 4;
  if (hipFuncSetAttribute((const void*)k<HP, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) printf("LDS %zu refused\n", lds);
  hipFuncSetAttribute((const void*)k<HP, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<HP, V>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, cyc, gemms, N, tail, two);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (double)h[wv]);
  printf("%s HP=%d N=%d blocks=%d matrices=%d (%.1f MB) tail=%d sources=%d%s: %.0f cycles, %.3f us per GEMM call (%d matri%s)\n",
         V ? "node_gemm_h (fp16 pairs)" : "node_gemm   (fp32 MFMA) ", HP, N, blocks, nmat, (double)nmat * T * T * 1024 / 1e6, tail, two ? 2 : 1,
         two == 2 ? " split in turn" : "", mx / gemms, ms * 1e3 / gemms, two ? 2 : 1, two ? "ces" : "x");
  if (GAUDI_NODE_STAMPS && V == 3) {
    std::vector<unsigned long long> hs(64);
    hipMemcpy(hs.data(), cyc + blocks * 8, 64 * 8, hipMemcpyDeviceToHost);
    printf("    cycles per call by part: prologue loads | split | barrier | init+B | K loop | fold | epilogue | closing barrier\n");
    for (int wv : {0, 3, 4, 7}) {
      printf("    wave %d:", wv);
      for (int i = 0; i < 8; ++i) printf(" %6.0f", (double)hs[wv * 8 + i] / gemms);
      printf("\n");
    }
  }
  hipFree(out);
  hipFree(cyc);
  hipFree(w);
}

// amp: 0 = unit normal activations; 1 = rows scaled by 10^(+-8) (far beyond fp16's range both ways); 2 = one huge entry per row
// beside tiny ones
template <int HP, int V>
void run_num(int H, int N, int tail, int two, int amp = 0) {
  constexpr int T = HP / 16;
  std::mt19937 rng(11 + N);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> Wa((size_t)H * H), Wb((size_t)H * H), x((size_t)96 * HP, 0.f), bias(HP, 0.f);
  for (auto& v : Wa) v = nd(rng) / std::sqrt((float)H);
  for (auto& v : Wb) v = nd(rng) / std::sqrt((float)H);
  for (int n = 0; n < 96; ++n)
    for (int f = 0; f < H; ++f) {
      float v = nd(rng);
      if (amp == 1) v *= std::pow(10.f, (float)((n % 5) * 4 - 8));
      if (amp == 2) v *= (f == (n * 7) % H) ? 3e5f : 1e-5f;
      x[(size_t)n * HP + f] = v;
    }
  for (int f = 0; f < H; ++f) bias[f] = amp ? 0.f : nd(rng);
  float wmax = 0;
  for (auto v : Wa) wmax = std::max(wmax, std::fabs(v));
  for (auto v : Wb) wmax = std::max(wmax, std::fabs(v));
  int ex;
  std::frexp(wmax, &ex);  // wmax = f * 2^ex, f in [0.5, 1)
  const float scale = std::ldexp(1.f, 14 - ex), winv = std::ldexp(1.f, ex - 14);
  std::vector<float> pk((size_t)2 * T * T * 256 * (V == 3 ? 2 : 1), 0.f);
  if (V == 3) {
    pack_f16<HP>(pk.data(), Wa, H, tail != 0, scale);
    pack_f16<HP>(pk.data() + (size_t)2 * T * T * 256, Wb, H, tail != 0, scale);
  } else {
    pack_f32<HP>(pk.data(), Wa, H);
    pack_f32<HP>(pk.data() + (size_t)T * T * 256, Wb, H);
  }
  float *dw, *dx, *db, *dy;
  hipMalloc(&dw, pk.size() * 4);
  hipMalloc(&dx, x.size() * 4);
  hipMalloc(&db, HP * 4);
  hipMalloc(&dy, (size_t)N * HP * 4);
  hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), HP * 4, hipMemcpyHostToDevice);
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  const size_t lds = (3 * N * (HP + 4) + ((HP + 63) / 64) * 64 + (two == 1 ? 2 : 1) * w8::nh_split_floats(HP, nct) + 96) * 4;
  if (hipFuncSetAttribute((const void*)k_num<HP, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) printf("LDS %zu refused\n", lds);
  hipLaunchKernelGGL((k_num<HP, V>), dim3(1), dim3(512), lds, 0, dw, (unsigned)(pk.size() * 4), dx, db, dy, N, tail, two, winv);
  std::vector<float> y((size_t)N * HP);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed (LDS %zu)\n", lds);
  hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
  // error per ROW relative to the row's largest |reference| (rows differ by 16 orders of magnitude in the amp cases)
  double worst = 0, padmax = 0;
  int nan = 0;
  for (int n = 0; n < N; ++n) {
    double emax = 0, rmax = 0;
    for (int o = 0; o < HP; ++o) {
      const float got = y[(size_t)n * HP + o];
      if (got != got) { ++nan; continue; }
      if (o >= H) { padmax = std::max(padmax, (double)std::fabs(got)); continue; }
      double s = bias[o];
      for (int kk = 0; kk < H; ++kk) {
        s += (double)Wa[(size_t)o * H + kk] * x[(size_t)n * HP + kk];
        if (two) s += (double)Wb[(size_t)o * H + kk] * x[(size_t)(48 + n) * HP + kk];
      }
      emax = std::max(emax, std::fabs(s - got));
      rmax = std::max(rmax, std::fabs(s));
    }
    worst = std::max(worst, emax / rmax);
  }
  printf("numerics %s H=%d HP=%d N=%d sources=%d%s tail=%d amp=%d: worst row max|err| / max|ref| vs float64 %.2e; padding max %.1e; NaN %d  %s\n",
         V ? "node_gemm_h" : "node_gemm  ", H, HP, N, two ? 2 : 1, two == 2 ? "(in turn)" : "", tail, amp, worst, padmax, nan,
         (worst < 2e-6 && padmax == 0 && nan == 0) ? "OK" : "FAIL");
  hipFree(dw); hipFree(dx); hipFree(db); hipFree(dy);
}

template <int V>
void numerics() {
  for (int tail = 0; tail < 2; ++tail) {
    run_num<208, V>(196, 11, tail, 0);
    run_num<208, V>(196, 11, tail, 1);
    run_num<208, V>(196, 16, tail, 2);
    run_num<48, V>(36, 7, tail, 1);
    run_num<48, V>(36, 3, tail, 0);
  }
  run_num<208, V>(196, 11, 1, 1, 1);
  run_num<208, V>(196, 11, 1, 0, 2);
  run_num<192, V>(192, 16, 0, 2, 1);
  if (V == 0 || GAUDI_MB_MAXNT >= 2) run_num<208, V>(196, 22, 1, 1);
  if (V == 0 || GAUDI_MB_MAXNT >= 3) {
    run_num<208, V>(196, 40, 1, 2);
    run_num<192, V>(192, 40, 0, 2, 2);
  }
  run_num<208, V>(208, 11, 0, 1);
  run_num<192, V>(192, 11, 0, 0);
  run_num<192, V>(192, 16, 0, 1);
  run_num<32, V>(32, 5, 0, 1);
  run_num<64, V>(64, 9, 0, 0);
  run_num<128, V>(128, 12, 0, 2);
  run_num<256, V>(256, 11, 0, 1);
}
template <int V>
void timing() {
  run<192, V>(11, 256, 63, 0);
  run<208, V>(11, 256, 120, 1);
  run<208, V>(11, 1, 120, 1);
  run<208, V>(16, 256, 120, 1);
  run<208, V>(11, 256, 1, 1);  // the weights resident in L2 (one matrix)
  run<208, V>(11, 256, 120, 1, 1);  // two sources per call, both split copies up front
  run<208, V>(11, 256, 120, 1, 2);  // ... split in turn (one region)
  if (V == 0 || GAUDI_MB_MAXNT >= 2) {
    run<208, V>(20, 256, 120, 1);     // two column tiles (hetero molecules)
    run<208, V>(20, 256, 120, 1, 2);
  }
  if (V == 0 || GAUDI_MB_MAXNT >= 3) run<208, V>(40, 256, 120, 1);     // three column tiles
}

int main(int argc, char**) {
  printf("depth %d, column tiles <= %d, ablation %d\n", GAUDI_NODE_DEPTH, GAUDI_MB_MAXNT, GAUDI_NODE_ABLATE);
  if (argc > 1) {  // any argument: the fp16 form's timings only
    timing<3>();
    return 0;
  }
  numerics<0>();
  timing<0>();
  numerics<3>();
  timing<3>();
  return 0;
}
