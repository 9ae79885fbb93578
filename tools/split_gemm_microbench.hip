// Microbenchmark + numerics probe of the split edge GEMM (w8_split.h: fp16 pairs since round 5; rounds 2-4 measured the bf16 x 3
// form with this file, profiles/r02d_split_gemm_microbench.txt) next to the fp32-MFMA one (w8_common.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I gaudi_amd/csrc tools/split_gemm_microbench.hip -o gaudi_amd/split_mb && gaudi_amd/split_mb
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "w8_split.h"
using namespace gaudi;
#ifndef SPLIT_MODE
#define SPLIT_MODE 1  // 2: half ring
#endif

static uint16_t f16_rne(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static float f16_f(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
constexpr float kWScale = 16384.f;  // 2^14: the weights below are < 1 in magnitude
// W[o][k] (H x H, row-major) -> split units [m][t][p] of W * kWScale
template <int HP>
static void pack_split(std::vector<float>& dst, const std::vector<float>& W, int H) {
  using G = w8::SplitGeo<HP, SPLIT_MODE>;
  dst.assign(G::kMatFloats, 0.f);
  uint16_t* d = (uint16_t*)dst.data();
  for (int m = 0; m < G::NC; ++m)
    for (int t = 0; t < G::T; ++t)
      for (int L = 0; L < 64; ++L)
        for (int e = 0; e < 8; ++e) {
          const int row = L & 15, g = L >> 4, tile = 2 * m + (e >> 2), k = 16 * tile + 4 * g + (e & 3), o = 16 * t + row;
          const float w = ((tile < G::T && k < H && o < H) ? W[(size_t)o * H + k] : 0.f) * kWScale;
          const uint16_t hi = f16_rne(w);
          d[((size_t)((m * G::T + t) * 2 + 0) * 64 + L) * 8 + e] = hi;
          d[((size_t)((m * G::T + t) * 2 + 1) * 64 + L) * 8 + e] = f16_rne(w - f16_f(hi));
        }
}
// lane-linear fp32 tiles [k/16][o/16], float4 index L = (row L & 15, k-quad L >> 4)
template <int HP>
static void pack_f32(std::vector<float>& dst, const std::vector<float>& W, int H) {
  constexpr int T = HP / 16;
  dst.assign((size_t)T * T * 256, 0.f);
  for (int kc = 0; kc < T; ++kc)
    for (int t = 0; t < T; ++t)
      for (int L = 0; L < 64; ++L)
        for (int q = 0; q < 4; ++q) {
          const int o = 16 * t + (L & 15), k = 16 * kc + 4 * (L >> 4) + q;
          dst[((size_t)(kc * T + t) * 64 + L) * 4 + q] = (o < H && k < H) ? W[(size_t)o * H + k] : 0.f;
        }
}

template <int HP, bool SPLIT>
__global__ __launch_bounds__(512) void k_time(const float* w, unsigned wbytes, int nmat, float* out, unsigned long long* cyc,
                                              int gemms, int nactive) {
  constexpr int T = HP / 16, LD = HP + 4, N = 11;
  using G = w8::SplitGeo<HP, SPLIT_MODE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ring_floats = SPLIT ? w8::EdgeRing<HP, SPLIT_MODE>::kFloats : 2 * T * 256;
  float* sP = smem + ring_floats;
  float* sQ = sP + N * LD;
  float* vec = sQ + N * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
  for (int i = tid; i < 2 * N * LD + 3 * HP; i += 512) sP[i] = 0.01f * ((i * 7) % 13) - 0.05f;
  __syncthreads();
  const WBuf wb = make_wbuf(w, wbytes);
  const bool active = wave < nactive;
  const int i = (wave + c) % N, j = (wave * 3 + c) % N;
  f4 total = splat(0.f);
  constexpr int MF = SPLIT ? G::kMatFloats : T * T * 256;
  w8::Ring<HP> ring;
  w8::RingS<HP, SPLIT_MODE> rs;
  if (SPLIT) {
    w8::er_init(rs, smem, false, w, 1.0f / kWScale);
    w8::rings_start(rs, wb, 0, wave, lane);
  } else {
    ring.base = smem;
    ring.par = 0;
    ring.ktail = false;
    w8::ring_start<HP>(ring, wb, 0, wave, lane);
  }
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * MF, nextW = ((m + 1) % nmat) * MF;
    f4 acc[T];
    if (SPLIT)
      w8::edge_gemm_pq_s(acc, rs, wb, W, nextW, vec, vec + HP, vec + 2 * HP, sP + i * LD + 4 * g, sQ + j * LD + 4 * g, 0.3f,
                             0.7f, 0.5f, active, wave, lane);
    else
      w8::edge_gemm_pq<HP>(acc, ring, wb, W, nextW, vec, vec + HP, vec + 2 * HP, sP + i * LD + 4 * g, sQ + j * LD + 4 * g, 0.3f,
                           0.7f, active, wave, lane);
#pragma unroll
    for (int t = 0; t < T; ++t) total = total + acc[t];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = total[0] + total[1] + total[2] + total[3];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (tid == 0 && blockIdx.x == 0) cyc[gridDim.x * 8] = wall_clock64() - w0;
}

template <int HP, bool SPLIT>
static size_t ring_bytes();
// ---- probe for a "one ring pass for two rounds of edge tiles" kernel (wide groups run their two rounds one
// after the other, each with its own pass over the weight ring): the chained split GEMM with TWO output tiles sets per wave and trip --
// every A unit is read from LDS once and multiplied into both sets, the ring traffic and the trip barriers are shared.  NT2 = 1 is the
// production form (w8::edge_gemm_regs_s), NT2 = 2 the probe; the second set multiplies the same B operand (what is timed is the
// trip structure, not the numerics).
template <int HP, int NT, int NT2, class MID>
__device__ __forceinline__ void rings_mfma_act2(f4* acc0, f4* acc1, const float* slot_lane, const w8::B3& b0, const w8::B3& b1, MID mid) {
  constexpr int U = w8::SplitGeo<HP, SPLIT_MODE>::kUnit;
  constexpr int kAhead = NT > 2 ? 2 : 1, kSets = kAhead + 1;
  f4 a[kSets][w8::kPieces];
#pragma unroll
  for (int q = 0; q < kAhead; ++q)
#pragma unroll
    for (int p = 0; p < w8::kPieces; ++p) a[q][p] = *(const f4*)(slot_lane + (q * w8::kPieces + p) * U);
  w8::static_for<NT>([&](auto t_tag) {
    constexpr int t = decltype(t_tag)::value;
    constexpr int cur = t % kSets;
    if (t == NT / 2) {
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + kAhead < NT) {
#pragma unroll
      for (int p = 0; p < w8::kPieces; ++p) a[(t + kAhead) % kSets][p] = *(const f4*)(slot_lane + ((t + kAhead) * w8::kPieces + p) * U);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const w8::u4 ah = __builtin_bit_cast(w8::u4, a[cur][0]), al = __builtin_bit_cast(w8::u4, a[cur][1]);
      f4 c0 = acc0[t];
      c0 = w8::mfma_bf(al, b0.h, c0);
      c0 = w8::mfma_bf(ah, b0.l, c0);
      c0 = w8::mfma_bf(ah, b0.h, c0);
      acc0[t] = c0;
      if constexpr (NT2 == 2) {
        f4 c1 = acc1[t];
        c1 = w8::mfma_bf(al, b1.h, c1);
        c1 = w8::mfma_bf(ah, b1.l, c1);
        c1 = w8::mfma_bf(ah, b1.h, c1);
        acc1[t] = c1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}
template <int HP, int NT2>
__global__ __launch_bounds__(512) void k_time2(const float* w, unsigned wbytes, int nmat, float* out, int gemms, int nactive) {
  constexpr int T = HP / 16;
  using G = w8::SplitGeo<HP, SPLIT_MODE>;
  static_assert(!(G::kTailOK && false), "");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WBuf wb = make_wbuf(w, wbytes);
  w8::RingS<HP, SPLIT_MODE> rs;
  w8::er_init(rs, smem, false, w, 1.0f / kWScale);
  w8::rings_start(rs, wb, 0, wave, lane);
  f4 in[T], o0[T], o1[T];
#pragma unroll
  for (int t = 0; t < T; ++t) in[t] = (f4){0.01f * (lane + t), 0.02f * t, 0.5f, -0.25f};
  f4 total = splat(0.f);
  constexpr int MF = G::kMatFloats;
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * MF, nextW = ((m + 1) % nmat) * MF;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      o0[t] = splat(0.f);
      o1[t] = splat(0.f);
    }
    w8::static_for<G::NC>([&](auto m_tag) {
      constexpr int mc = decltype(m_tag)::value;
      const w8::B3 bin = w8::split8(in[2 * mc < T ? 2 * mc : 0], in[2 * mc + 1 < T ? 2 * mc + 1 : 0], 1.0f);
      // (the second set's operand: other registers, so that the two sets are two computations)
      const w8::B3 bin1 = w8::split8(in[(2 * mc + 3) % T], in[(2 * mc + 5) % T], 0.5f);
      w8::static_for<G::NH>([&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
        constexpr int tr = mc * G::NH + h;
        w8::trip_open(rs, lane);
        if (wave < nactive)
          rings_mfma_act2<HP, G::tiles_of(h), NT2>(o0 + h * G::CH, o1 + h * G::CH, rs.slot(rs.par) + lane * 4, bin, bin1,
                                                   [&] { w8::rings_stage(rs, wb, W, nextW, tr, wave, lane); });
        else  // a wave without an edge tile: its share of the ring traffic only
          w8::rings_stage(rs, wb, W, nextW, tr, wave, lane);
        w8::trip_close(rs, lane);
      });
    });
#pragma unroll
    for (int t = 0; t < T; ++t) {
      total = total + o0[t] + o1[t];
      in[t] = in[t] + o0[t] * 1e-30f + o1[(t + 1) % T] * 1e-30f;
    }
  }
  out[blockIdx.x * 512 + tid] = total[0] + total[1] + total[2] + total[3];
}
template <int HP, int NT2>
void run_time2(int blocks, int nmat, int nactive = 8) {
  using G = w8::SplitGeo<HP, SPLIT_MODE>;
  const size_t wfloats = (size_t)G::kMatFloats * nmat;
  float *out, *w;
  hipMalloc(&w, wfloats * 4);
  {
    std::vector<uint16_t> h(wfloats * 2);
    std::mt19937 rng(1);
    for (auto& v : h) v = f16_rne(0.1f * ((int)(rng() % 2001) - 1000) / 1000.f);
    hipMemcpy(w, h.data(), wfloats * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&out, blocks * 512 * 4);
  const int gemms = 200;
  const size_t lds = ring_bytes<HP, true>();
  hipFuncSetAttribute((const void*)k_time2<HP, NT2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_time2<HP, NT2>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, gemms, nactive);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  printf("chained split GEMM, %d waves x %d output-tile set(s) per trip (%d edge tiles), HP=%d blocks=%d matrices=%d: %.2f us per GEMM (K=%d) = %.2f us per edge tile\n",
         nactive, NT2, nactive * NT2, HP, blocks, nmat, ms * 1e3 / gemms, HP, ms * 1e3 / gemms / (NT2 * nactive));
  hipFree(out);
  hipFree(w);
}

// numerics: out[e][o] = sum_k W[o][k] in[e][k] for 128 edge columns (8 waves x 16), chained-GEMM form
template <int HP, bool SPLIT>
__global__ __launch_bounds__(512) void k_num(const float* w, unsigned wbytes, const float* in, float* out) {
  constexpr int T = HP / 16;
  using G = w8::SplitGeo<HP, SPLIT_MODE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
  const WBuf wb = make_wbuf(w, wbytes);
  f4 x[T], y[T];
  const int e = wave * 16 + c;
#pragma unroll
  for (int t = 0; t < T; ++t) x[t] = *(const f4*)(in + (size_t)e * HP + 16 * t + 4 * g);
  if (SPLIT) {
    w8::RingS<HP, SPLIT_MODE> rs;
    w8::er_init(rs, smem, false, w, 1.0f / kWScale);
    w8::rings_start(rs, wb, 0, wave, lane);
    w8::edge_gemm_regs_s(y, x, rs, wb, 0, -1, nullptr, nullptr, true, wave, lane);
  } else {
    w8::Ring<HP> ring;
    ring.base = smem;
    ring.par = 0;
    ring.ktail = false;
    w8::ring_start<HP>(ring, wb, 0, wave, lane);
    w8::edge_gemm_regs<HP>(y, x, ring, wb, 0, -1, nullptr, nullptr, true, wave, lane);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) *(f4*)(out + (size_t)e * HP + 16 * t + 4 * g) = y[t];
}

template <int HP, bool SPLIT>
static size_t ring_bytes() {
  return (SPLIT ? w8::EdgeRing<HP, SPLIT_MODE>::kFloats : 2 * (HP / 16) * 256) * 4;
}

template <int HP, bool SPLIT>
void run_time(int nactive, int blocks, int nmat) {
  constexpr int T = HP / 16;
  using G = w8::SplitGeo<HP, SPLIT_MODE>;
  const size_t mf = SPLIT ? G::kMatFloats : T * T * 256, wfloats = mf * nmat;
  float *out, *w;
  unsigned long long* cyc;
  hipMalloc(&w, wfloats * 4);
  {
    std::vector<uint16_t> h(wfloats * 2);
    std::mt19937 rng(1);
    for (auto& v : h) v = SPLIT ? f16_rne(0.1f * ((int)(rng() % 2001) - 1000) / 1000.f) : 0;
    if (!SPLIT) {
      float* f = (float*)h.data();
      for (size_t i = 0; i < wfloats; ++i) f[i] = 0.1f * ((int)(rng() % 2001) - 1000) / 1000.f;
    }
    hipMemcpy(w, h.data(), wfloats * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8 + 8);
  const int gemms = 200;
  const size_t lds = ring_bytes<HP, SPLIT>() + (2 * 11 * (HP + 4) + 3 * HP) * 4;
  hipFuncSetAttribute((const void*)k_time<HP, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_time<HP, SPLIT>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, cyc, gemms,
                       nactive);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  unsigned long long hc[2];
  hipMemcpy(&hc[0], cyc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&hc[1], cyc + blocks * 8, 8, hipMemcpyDeviceToHost);
  {  // checksum of every thread's accumulated outputs: equal between the barrier and the flag form, and from run to run
    std::vector<float> ho((size_t)blocks * 512);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    uint64_t hsh = 1469598103934665603ull;
    for (float v : ho) {
      uint32_t u;
      memcpy(&u, &v, 4);
      hsh = (hsh ^ u) * 1099511628211ull;
    }
    double sum = 0, asum = 0;
    for (float v : ho) { sum += v; asum += std::fabs(v); }
    printf("[out hash %016llx sum %.9e abs %.9e] ", (unsigned long long)hsh, sum, asum);
  }
  printf("[memtime %llu ticks, wall %llu ticks of 100 MHz => memtime runs at %.0f MHz] ", hc[0], hc[1], 100.0 * hc[0] / hc[1]);
  printf("%s edge_gemm_pq HP=%d CH=%d active=%d blocks=%d matrices=%d (%.1f MB) LDS %zu B: %.2f us per GEMM (K=%d)\n",
         SPLIT ? "fp16 pairs" : "fp32-mfma ", HP, SPLIT ? G::CH : 0, nactive, blocks, nmat, wfloats * 4 / 1e6, lds, ms * 1e3 / gemms, HP);
  hipFree(out);
  hipFree(cyc);
  hipFree(w);
}

template <int HP>
void run_num(int H) {
  constexpr int E = 128;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> W((size_t)H * H), in((size_t)E * HP, 0.f);
  for (auto& v : W) v = nd(rng) / std::sqrt((float)H);
  for (int e = 0; e < E; ++e)
    for (int k = 0; k < H; ++k) in[(size_t)e * HP + k] = nd(rng) * (1.f + (e % 5));
  std::vector<double> ref((size_t)E * H);
  std::vector<float> ref32((size_t)E * H);
  double refmax = 0;
  for (int e = 0; e < E; ++e)
    for (int o = 0; o < H; ++o) {
      double s = 0;
      float s32 = 0.f;
      for (int k = 0; k < H; ++k) {
        s += (double)W[(size_t)o * H + k] * in[(size_t)e * HP + k];
        s32 = fmaf(W[(size_t)o * H + k], in[(size_t)e * HP + k], s32);
      }
      ref[(size_t)e * H + o] = s;
      ref32[(size_t)e * H + o] = s32;
      refmax = std::max(refmax, std::fabs(s));
    }
  float *din, *dout, *dw;
  hipMalloc(&din, in.size() * 4);
  hipMalloc(&dout, in.size() * 4);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  for (int split = 0; split < 2; ++split) {
    std::vector<float> pk;
    if (split) pack_split<HP>(pk, W, H);
    else pack_f32<HP>(pk, W, H);
    hipMalloc(&dw, pk.size() * 4);
    hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = split ? ring_bytes<HP, true>() : ring_bytes<HP, false>();
    if (split) {
      hipFuncSetAttribute((const void*)k_num<HP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_num<HP, true>), dim3(1), dim3(512), lds, 0, dw, (unsigned)(pk.size() * 4), din, dout);
    } else {
      hipFuncSetAttribute((const void*)k_num<HP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_num<HP, false>), dim3(1), dim3(512), lds, 0, dw, (unsigned)(pk.size() * 4), din, dout);
    }
    std::vector<float> got(in.size());
    hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, erms = 0, e32 = 0;
    for (int e = 0; e < E; ++e)
      for (int o = 0; o < H; ++o) {
        const double d = got[(size_t)e * HP + o] - ref[(size_t)e * H + o];
        emax = std::max(emax, std::fabs(d));
        erms += d * d;
        e32 = std::max(e32, std::fabs((double)ref32[(size_t)e * H + o] - ref[(size_t)e * H + o]));
      }
    printf("numerics H=%d HP=%d %s: max|err| vs float64 = %.3e (rms %.3e) of max|ref| %.3e -> %.2e relative; host fmaf chain: %.2e\n", H, HP,
           split ? "fp16 pairs x3" : "fp32 MFMA    ", emax, std::sqrt(erms / (E * H)), refmax, emax / refmax, e32 / refmax);
    hipFree(dw);
  }
  hipFree(din);
  hipFree(dout);
}

int main(int argc, char** argv) {
  if (argc > 1) {
    run_time<208, true>(7, 256, 48);
    run_time<208, true>(7, 1, 48);
    return 0;
  }
  run_num<192>(192);
  run_num<208>(196);
  run_num<48>(36);
  run_time<192, false>(7, 256, 27);
  run_time<192, true>(7, 256, 27);
  run_time<192, true>(8, 256, 27);
  run_time<208, false>(7, 256, 48);
  run_time<208, true>(7, 256, 48);
  run_time<208, true>(7, 1, 48);
  for (int na : {7, 8, 4}) {
    run_time2<192, 1>(256, 27, na);
    run_time2<192, 2>(256, 27, na);
    run_time2<208, 1>(256, 48, na);
    run_time2<208, 2>(256, 48, na);
  }
  return 0;
}
