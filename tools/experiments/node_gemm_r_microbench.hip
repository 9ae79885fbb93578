// Microbenchmark + numerics probe of the ROLE-SPLIT node GEMM (w8_nodes_role.h: node_gemm_r -- waves 4-7 stream the fp16-pair
// images by LDS-DMA into per-pair LDS rings, waves 0-3 multiply) against the form it would replace (w8_nodes_f16.h: node_gemm_h,
// every wave loads and multiplies).  Same images, same accumulation order: the outputs must agree bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I gaudi_amd/csrc -I tools/experiments tools/experiments/node_gemm_r_microbench.hip -o gaudi_amd/ngemmr_mb
//   -DGAUDI_ROLE_RING=<KiB per pair> -DGAUDI_ROLE_FLIGHT=<loads in flight> -DGAUDI_ROLE_ABLATE=<1|2> -DGAUDI_NODE_ABLATE=<1|2>
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "w8_nodes_role.h"
using namespace gaudi;
#ifndef GAUDI_MB_MAXNT
#define GAUDI_MB_MAXNT 2
#endif
constexpr int kR = GAUDI_ROLE_RING, kF = GAUDI_ROLE_FLIGHT;
constexpr int kRA = (GAUDI_ROLE_ABLATE & 1) && kR > 12 ? 12 : kR;  // slots allocated

// timing: `gemms` calls over a set of nmat matrices; phase > 0: every `phase` calls the chain ends (nextW = -1, the loaders flush,
// a full barrier) and restarts cold -- a node phase between two edge phases of the sampler
template <int HP, int ROLE>
__global__ __launch_bounds__(512) void k(const float* w, unsigned wbytes, int nmat, float* out, unsigned long long* cyc, int gemms, int N, int tail,
                                         int two, int phase) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Rw = (N + 15) & ~15;
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  float* sX = smem;
  float* sY = sX + Rw * LD;
  float* sSplit = smem + 2 * Rw * LD;
  float* sRing = sSplit + 96 + 2 * w8::nh_split_floats(HP, nct);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * Rw * LD; i += 512) sX[i] = 0.01f * ((i * 7) % 13) - 0.05f;
  const WBuf wb = make_wbuf(w, wbytes);
  const int MS = T * T * 256;
  w8::NodeCtxH cx{1.0f, sSplit + 96, two == 2 ? sSplit + 96 : sSplit + 96 + w8::nh_split_floats(HP, nct), tail != 0, sSplit};
  w8::RoleState rs;
  w8::NodePFH<HP> pf;
  if (ROLE) w8::role_init<kRA>(rs, sRing, wave, tid);
  else w8::node_prefetch_h<HP>(pf, wb, 0, wave, lane);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int m = 0; m < gemms; ++m) {
    const int W = (m % nmat) * MS;
    const bool last = phase > 0 && (m + 1) % phase == 0;
    const int nextW = last ? -1 : ((m + 1) % nmat) * MS;
    const float* X = (m & 1) ? sY : sX;
    float* Y = (m & 1) ? sX : sY;
    const int W2 = ((m + nmat / 2) % nmat) * MS;
    if (ROLE) {
      if (two) w8::node_gemm_r<HP, EPI_SILU, true, GAUDI_MB_MAXNT, kR, kF>(rs, w, W, X, true, W2, X, nullptr, Y, nullptr, nullptr, N, wave, lane, cx, nextW);
      else w8::node_gemm_r<HP, EPI_SILU, false, GAUDI_MB_MAXNT, kR, kF>(rs, w, W, X, true, -1, nullptr, nullptr, Y, nullptr, nullptr, N, wave, lane, cx, nextW);
      if (last) __syncthreads();
      else w8::lds_barrier();
    } else {
      if (two) w8::node_gemm_h<HP, EPI_SILU, true, GAUDI_MB_MAXNT, w8::kAheadAll, w8::kAheadAll>(wb, W, X, true, W2, X, nullptr, Y, nullptr, nullptr, N, wave, lane, cx, pf, nextW);
      else w8::node_gemm_h<HP, EPI_SILU, false, GAUDI_MB_MAXNT, w8::kAheadAll, w8::kAheadAll>(wb, W, X, true, -1, nullptr, nullptr, Y, nullptr, nullptr, N, wave, lane, cx, pf, nextW);
      __syncthreads();
      if (last) w8::node_prefetch_h<HP>(pf, wb, ((m + 1) % nmat) * MS, wave, lane);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + tid] = sX[tid % (Rw * LD)];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// numerics: Y[n][o] = bias[o] + sum_k Wa[o][k] Xa[n][k] + sum_k Wb[o][k] Xb[n][k]   (one workgroup, EPI_NONE, `calls` calls in a row
// on the same data so that the ring wraps and the cross-call prefill is exercised: every call must give the same result)
template <int HP, int ROLE>
__global__ __launch_bounds__(512) void k_num(const float* w, unsigned wbytes, const float* x, const float* bias, float* y, int N, int tail, int two,
                                             float winv, int calls) {
  constexpr int T = HP / 16, LD = HP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sXa = smem;
  float* sXb = sXa + N * LD;
  float* sY = sXb + N * LD;
  float* sB = sY + N * LD;
  float* sSplit = sB + ((HP + 63) / 64) * 64;
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  float* sRing = sSplit + 96 + 2 * w8::nh_split_floats(HP, nct);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < N * LD; i += 512) {
    const int n = i / LD, f = i % LD;
    sXa[i] = (n < N && f < HP) ? x[n * HP + f] : 0.f;
    sXb[i] = (n < N && f < HP) ? x[(48 + n) * HP + f] : 0.f;
    sY[i] = __builtin_nanf("");
  }
  for (int i = tid; i < HP; i += 512) sB[i] = bias[i];
  for (int i = tid; i < 2 * w8::nh_split_floats(HP, nct) + 96; i += 512) sSplit[i] = __builtin_nanf("");
  const WBuf wb = make_wbuf(w, wbytes);
  w8::NodeCtxH cx{winv, sSplit + 96, two == 2 ? sSplit + 96 : sSplit + 96 + w8::nh_split_floats(HP, nct), tail != 0, sSplit};
  w8::RoleState rs;
  w8::NodePFH<HP> pf;
  if (ROLE) w8::role_init<kRA>(rs, sRing, wave, tid);
  else w8::node_prefetch_h<HP>(pf, wb, 0, wave, lane);
  __syncthreads();
  const int Wb = two ? T * T * 256 : -1;
  for (int it = 0; it < calls; ++it) {
    const int nextW = it + 1 < calls ? 0 : -1;
    if (it > 0) {
      for (int i = tid; i < N * LD; i += 512) sY[i] = __builtin_nanf("");
      w8::lds_barrier();
    }
    if (ROLE) {
      if (two) w8::node_gemm_r<HP, EPI_NONE, true, GAUDI_MB_MAXNT, kR, kF>(rs, w, 0, sXa, true, Wb, sXb, sB, sY, nullptr, nullptr, N, wave, lane, cx, nextW);
      else w8::node_gemm_r<HP, EPI_NONE, false, GAUDI_MB_MAXNT, kR, kF>(rs, w, 0, sXa, true, -1, nullptr, sB, sY, nullptr, nullptr, N, wave, lane, cx, nextW);
    } else {
      if (two) w8::node_gemm_h<HP, EPI_NONE, true, GAUDI_MB_MAXNT, w8::kAheadAll, w8::kAheadAll>(wb, 0, sXa, true, Wb, sXb, sB, sY, nullptr, nullptr, N, wave, lane, cx, pf, nextW);
      else w8::node_gemm_h<HP, EPI_NONE, false, GAUDI_MB_MAXNT, w8::kAheadAll, w8::kAheadAll>(wb, 0, sXa, true, -1, nullptr, sB, sY, nullptr, nullptr, N, wave, lane, cx, pf, nextW);
    }
    w8::lds_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
// the fp16-pair image (w8_nodes_f16.h; gaudi_hip.hip: pack_matrix_f16)
template <int HP>
static void pack_f16(float* dst, const std::vector<float>& W, int H, float scale) {
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

static size_t lds_timing(int HP, int N) {
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  return (size_t)(2 * ((N + 15) & ~15) * (HP + 4) + 96 + 2 * w8::nh_split_floats(HP, nct) + w8::role_ring_floats(kRA)) * 4;
}

template <int HP, int ROLE>
double run(int N, int blocks, int nmat, int tail, int two = 0, int phase = 0) {
  constexpr int T = HP / 16;
  float *out, *w;
  unsigned long long* cyc;
  const size_t wfloats = (size_t)nmat * T * T * 256 * 2;
  hipMalloc(&w, wfloats * 4);
  hipMemset(w, 0, wfloats * 4);
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&cyc, blocks * 8 * 8);
  const int gemms = 600;
  const size_t lds = lds_timing(HP, N);
  if (hipFuncSetAttribute((const void*)k<HP, ROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) printf("LDS %zu refused\n", lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<HP, ROLE>), dim3(blocks), dim3(512), lds, 0, w, (unsigned)(wfloats * 4), nmat, out, cyc, gemms, N, tail, two, phase);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) printf("launch failed\n");
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (double)h[wv]);
  printf("%s HP=%d N=%d blocks=%d matrices=%d tail=%d sources=%d%s phase=%d LDS=%zu: %.0f cycles, %.3f us per GEMM call (%d matri%s)\n",
         ROLE ? "node_gemm_r (role split)" : "node_gemm_h (all waves) ", HP, N, blocks, nmat, tail, two ? 2 : 1, two == 2 ? " split in turn" : "", phase, lds,
         mx / gemms, ms * 1e3 / gemms, two ? 2 : 1, two ? "ces" : "x");
  fflush(stdout);
  hipFree(out);
  hipFree(cyc);
  hipFree(w);
  return ms * 1e3 / gemms;
}

template <int HP>
void run_num(int H, int N, int tail, int two, int amp = 0, int calls = 3) {
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
  std::frexp(wmax, &ex);
  const float scale = std::ldexp(1.f, 14 - ex), winv = std::ldexp(1.f, ex - 14);
  std::vector<float> pk((size_t)2 * T * T * 256 * 2, 0.f);
  pack_f16<HP>(pk.data(), Wa, H, scale);
  pack_f16<HP>(pk.data() + (size_t)2 * T * T * 256, Wb, H, scale);
  float *dw, *dx, *db, *dy;
  hipMalloc(&dw, pk.size() * 4);
  hipMalloc(&dx, x.size() * 4);
  hipMalloc(&db, HP * 4);
  hipMalloc(&dy, (size_t)N * HP * 4);
  hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), HP * 4, hipMemcpyHostToDevice);
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  const size_t lds = (size_t)(3 * N * (HP + 4) + ((HP + 63) / 64) * 64 + 2 * w8::nh_split_floats(HP, nct) + 96 + w8::role_ring_floats(kRA)) * 4;
  std::vector<float> y[2];
  for (int role = 0; role < 2; ++role) {
    y[role].assign((size_t)N * HP, 0.f);
    hipMemset(dy, 0xff, (size_t)N * HP * 4);
    if (role) {
      hipFuncSetAttribute((const void*)k_num<HP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_num<HP, 1>), dim3(1), dim3(512), lds, 0, dw, (unsigned)(pk.size() * 4), dx, db, dy, N, tail, two, winv, calls);
    } else {
      hipFuncSetAttribute((const void*)k_num<HP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_num<HP, 0>), dim3(1), dim3(512), lds, 0, dw, (unsigned)(pk.size() * 4), dx, db, dy, N, tail, two, winv, calls);
    }
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed (LDS %zu)\n", lds);
    hipMemcpy(y[role].data(), dy, y[role].size() * 4, hipMemcpyDeviceToHost);
  }
  double worst = 0, padmax = 0;
  int nan = 0, diffbits = 0;
  for (size_t i = 0; i < y[0].size(); ++i) diffbits += std::memcmp(&y[0][i], &y[1][i], 4) != 0;
  for (int n = 0; n < N; ++n) {
    double emax = 0, rmax = 0;
    for (int o = 0; o < HP; ++o) {
      const float got = y[1][(size_t)n * HP + o];
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
  printf("numerics node_gemm_r H=%d HP=%d N=%d sources=%d%s tail=%d amp=%d calls=%d: worst row max|err| / max|ref| vs float64 %.2e; padding max %.1e; NaN %d; "
         "values that differ from node_gemm_h bit for bit: %d  %s\n",
         H, HP, N, two ? 2 : 1, two == 2 ? "(in turn)" : "", tail, amp, calls, worst, padmax, nan, diffbits,
         (worst < 2e-6 && padmax == 0 && nan == 0 && diffbits == 0) ? "OK" : "FAIL");
  fflush(stdout);
  hipFree(dw); hipFree(dx); hipFree(db); hipFree(dy);
}

void numerics() {
  for (int tail = 0; tail < 2; ++tail) {
    run_num<208>(196, 11, tail, 0);
    run_num<208>(196, 11, tail, 1);
    run_num<208>(196, 16, tail, 2);
    run_num<48>(36, 7, tail, 1);
    run_num<48>(36, 3, tail, 0);
  }
  run_num<208>(196, 11, 1, 1, 1);
  run_num<208>(196, 11, 1, 0, 2);
  run_num<192>(192, 16, 0, 2, 1);
  if (GAUDI_MB_MAXNT >= 2) run_num<208>(196, 22, 1, 1);
  run_num<208>(208, 11, 0, 1);
  run_num<192>(192, 11, 0, 0);
  run_num<192>(192, 16, 0, 1);
  run_num<32>(32, 5, 0, 1);
  run_num<64>(64, 9, 0, 0);
  run_num<128>(128, 12, 0, 2);
  run_num<256>(256, 11, 0, 1);
  run_num<208>(196, 11, 1, 0, 0, 1);  // one call: no prefill, flush at the end
  run_num<192>(192, 11, 0, 1, 0, 7);
}
template <int ROLE>
void timing() {
  run<192, ROLE>(11, 256, 63, 0);
  run<208, ROLE>(11, 256, 120, 1);
  run<208, ROLE>(11, 1, 120, 1);
  run<208, ROLE>(11, 256, 1, 1);        // the weights resident in L2 (one matrix)
  run<208, ROLE>(11, 256, 120, 1, 1);   // two sources per call
  run<208, ROLE>(11, 256, 120, 1, 2);   // ... split in turn
  run<208, ROLE>(11, 256, 120, 1, 0, 5);  // node phases of five matrices, cold start each
  run<208, ROLE>(11, 256, 120, 1, 0, 2);  // ... of two
  if (GAUDI_MB_MAXNT >= 2) {
    run<208, ROLE>(20, 256, 120, 1);
    run<208, ROLE>(22, 256, 120, 1, 1);
  }
}

int main(int argc, char** argv) {
  printf("ring %d KiB per pair, %d loads in flight, column tiles <= %d, role ablation %d, node ablation %d\n", kR, kF, GAUDI_MB_MAXNT, GAUDI_ROLE_ABLATE,
         GAUDI_NODE_ABLATE);
  if (GAUDI_ROLE_ABLATE == 0 && GAUDI_NODE_ABLATE == 0 && !(argc > 1 && argv[1][0] == 't')) numerics();
  if (!(argc > 1 && argv[1][0] == 'r')) timing<0>();
  timing<1>();
  return 0;
}
