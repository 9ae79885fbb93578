// w8_nodes_presplit.h -- EXPERIMENT (round 4, measured and rejected; tools/node_gemm_s_microbench.hip is its only user):
// node-level GEMMs of the full-ring 8-wave kernels on PRE-SPLIT weight images.
// Result (profiles/r04z_node_gemm_presplit_microbench.txt, 256 workgroups, weights streamed from a 41 MB set): 10 134 cycles per
// H = 196 -> 208 matrix against 8 355 for the fp32 form, 7 779 against 6 480 at H = 192; equal (8 057 / 8 148) only when one
// matrix stays in L2.  By the additive issue model the form could reach ~7 500 (matrix 2 376 + 72 KiB of loads x 60 + ~800), a
// 10 % gain on the node GEMMs at best; what it measures on top is memory latency at the GEMM's head and tail (three operand
// sets in flight do not cover an Infinity-Cache round trip at ~400 SIMD cycles per chunk, and carrying more sets across the
// phase boundaries is not affordable in registers).  The stream microbenchmark's 3.14 vs 3.97 us (profiles/r04z) was for 16
// tiles x 7 chunks per matrix, not the real 13 x 6.5.  Numerics are at the fp32 instruction's level (same file).
//
// Y[n][o] = epi( sum_k Wa[o][k] Xa[n][k] (+ sum_k Wb[o][k] Xb[n][k]) + bias[o] ) for the <= 16 node columns of a workgroup:
// the P / Q / node-MLP GEMMs of every layer and their transposes in the reverse pass (edm/egnn/egnn_new.py:59-73,
// edm/egnn_predictor/gcl.py:240-250) -- 183 matrices per guided step, each streamed from L2 once per workgroup.
//
// The fp32 form (w8_common.h: node_gemm, v_mfma_f32_16x16x4_f32) costs matrix time + ~0.64 x load time whatever the
// instruction order (profiles/r04o): 3.97 us per 16-tile H = 208 matrix.  Here the weights are the SAME split images the edge
// GEMMs use (three bf16 pieces, units of 1 KiB ordered [K chunk][output tile][piece], at twice the fp32 offset in the split
// buffer: gaudi_hip.hip: pack_matrix_split), loaded from L2 straight into registers -- a unit is used by exactly one wave -- and
// the six piece products run on v_mfma_f32_16x16x32_bf16: 1.5x the bytes, 2.67x less matrix time, and the sum is what the loads
// alone cost (profiles/r04z: 3.14 us for the same matrix with a 40 MB working set, 2.69 with the weights in L2).
// The activations are split ONCE per GEMM input by all threads into the idle slot of the weight ring (node_split_rows:
// [chunk][piece][lane] uint4 = 3 KiB per chunk), so a wave reads its B operand with three ds_read_b128 per chunk instead of
// splitting it 44 vector instructions at a time (the measured-and-rejected NG4 form, tools/experiments/w8_nodes_split.h).
// Accuracy: the split form's (w8_split.h header): at the fp32 instruction's own error level against float64.
// Accumulation order per output element: K chunks in order, six piece products smallest first, then the K tail's fp32 step.
#pragma once
#include "w8_split.h"

namespace gaudi {
namespace w8 {

// bf16 chunks of a split image (the K tail of an H % 16 == 4 width is a trailing group of T fp32 tiles: SplitGeo::kTailOK)
template <int HP>
__device__ __forceinline__ int ns_chunks(bool ktail) {
  constexpr int T = HP / 16;
  return (ktail && (T & 1) && T >= 3) ? (T - 1) / 2 : (T + 1) / 2;
}
// floats of the split-activation buffer of one GEMM input (16 node columns)
__host__ __device__ constexpr int ns_split_floats(int HP) { return ((HP / 16 + 1) / 2) * 3 * 256; }

template <int HP>
struct NodePFS {
  u4 a[2][3];  // K chunk 0 of the wave's (up to two) output tiles, three pieces each
};

__device__ __forceinline__ u4 ldu4(const WBuf& wbs, int off_floats, int lane) {
  return __builtin_bit_cast(u4, ldw4n(wbs, off_floats, lane));
}

template <int HP, int NTW>
__device__ __forceinline__ void node_prefetch_s_n(NodePFS<HP>& pf, const WBuf& wbs, int W /* fp32 float offset */, int wave, int lane) {
  constexpr int T = HP / 16;
#pragma unroll
  for (int u = 0; u < NTW; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) pf.a[u][p] = ldu4(wbs, 2 * W + ((wave + kWaves * u) * 3 + p) * 256, lane);
  (void)T;
}
template <int HP>
__device__ __forceinline__ void node_prefetch_s(NodePFS<HP>& pf, const WBuf& wbs, int W, int wave, int lane) {
  constexpr int T = HP / 16;
  if (wave + kWaves < T) node_prefetch_s_n<HP, 2>(pf, wbs, W, wave, lane);
  else if (wave < T) node_prefetch_s_n<HP, 1>(pf, wbs, W, wave, lane);
}

// all threads: the N (<= 16) rows of sX [N][HP+4] (LDS) -> split B operands of every bf16 chunk, dst [chunk][piece][lane] uint4.
// Columns past N repeat row N - 1 (their results are never stored).  The caller places a barrier before the GEMM reads dst.
template <int HP>
__device__ __forceinline__ void node_split_rows(float* dst, const float* sX, int N, bool ktail, int tid) {
  constexpr int T = HP / 16, LD = HP + 4;
  const int nc = ns_chunks<HP>(ktail);
  const int tiles = (ktail && (T & 1) && T >= 3) ? T - 1 : T;  // tiles the bf16 chunks cover
  for (int idx = tid; idx < nc * 64; idx += kThreads) {
    const int m = idx >> 6, lane = idx & 63, c = lane & 15, g = lane >> 4;
    const float* x = sX + (c < N ? c : N - 1) * LD + 32 * m + 4 * g;
    const f4 lo = *(const f4*)x;
    const f4 hi = 2 * m + 1 < tiles ? *(const f4*)(x + 16) : splat(0.f);
    const B3 b = split8(lo, hi);
    u4* d = (u4*)(dst + (size_t)m * 3 * 256) + lane;
    d[0] = b.h;
    d[64] = b.m;
    d[128] = b.l;
  }
}

// the A operands of one K chunk for the wave's (up to two) output tiles
struct NodeASet {
  u4 p[2][3];
};
template <int HP>
__device__ __forceinline__ void ns_load(NodeASet& s, const WBuf& wbs, int Wa, int Wb, int nc, int KT, int cc, int wave, int lane, bool two) {
  constexpr int T = HP / 16;
  const int k = cc < KT ? cc : KT - 1;  // clamped past the end: surplus loads are unused
  const int base = k < nc ? 2 * Wa + k * (T * 3 * 256) : 2 * Wb + (k - nc) * (T * 3 * 256);
#pragma unroll
  for (int p = 0; p < 3; ++p) s.p[0][p] = ldu4(wbs, base + (wave * 3 + p) * 256, lane);
  if (two) {
#pragma unroll
    for (int p = 0; p < 3; ++p) s.p[1][p] = ldu4(wbs, base + ((wave + kWaves) * 3 + p) * 256, lane);
  }
}
// LDS writes of every wave visible to every wave; global loads in flight stay in flight (__syncthreads would wait for them)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// K loop + epilogue of one wave: three operand sets in flight (a chunk's matrix instructions take ~400 SIMD cycles, an L2 /
// Infinity Cache round trip 2-3x that); S0 = chunk 0 (the cross-call prefetch), S1 / S2 = chunks 1 / 2 (loaded by node_gemm_s
// before the split barrier)
template <int HP, int EPI, int NTW>
__device__ __forceinline__ void node_gemm_s_body(const WBuf& wbs, int Wa, const float* sBa, const float* sXa, int Wb, const float* sBb,
                                                 const float* sXb, const float* sBias, float* sY, const float* sRes,
                                                 const float* sMask, int N, int wave, int lane, bool ktail, NodeASet& S0,
                                                 NodeASet& S1, NodeASet& S2, NodePFS<HP>* pf, int nextW, float* gPre) {
  constexpr int T = HP / 16, LD = HP + 4;
  const int c = lane & 15, g = lane >> 4;
  const int nc = ns_chunks<HP>(ktail);
  const bool tail = ktail && (T & 1) && T >= 3;
  const int KT = Wb >= 0 ? 2 * nc : nc;  // two sources run as ONE K loop so the load pipeline never restarts
  auto bld = [&](int cc) {
    const u4* q = (const u4*)((cc < nc ? sBa : sBb) + (size_t)(cc < nc ? cc : cc - nc) * 3 * 256) + lane;
    return (B3){q[0], q[64], q[128]};
  };
  f4 acc[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) acc[u] = sBias != nullptr ? *(const f4*)(sBias + 16 * (wave + kWaves * u) + 4 * g) : splat(0.f);
  auto mm = [&](const NodeASet& s, const B3& b) {
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      f4 y = acc[u];
      y = mfma_bf(s.p[u][2], b.h, y);  // small terms first (as the edge GEMMs: w8_split.h)
      y = mfma_bf(s.p[u][0], b.l, y);
      y = mfma_bf(s.p[u][1], b.m, y);
      y = mfma_bf(s.p[u][1], b.h, y);
      y = mfma_bf(s.p[u][0], b.m, y);
      y = mfma_bf(s.p[u][0], b.h, y);
      acc[u] = y;
    }
  };
  auto step = [&](NodeASet& s, int cc) {  // consume chunk cc from s, refill s with chunk cc + 3
    const B3 b = bld(cc);
    __builtin_amdgcn_sched_barrier(0);
    mm(s, b);
    __builtin_amdgcn_sched_barrier(0);
    if (cc + 3 < KT) ns_load<HP>(s, wbs, Wa, Wb, nc, KT, cc + 3, wave, lane, NTW == 2);
    __builtin_amdgcn_sched_barrier(0);
  };
  // the K tail's weights (one fp32 tile unit per output tile and source) travel with the chunks
  float ta[NTW], tb[NTW];
  if (tail) {
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      ta[u] = ldw4n(wbs, 2 * Wa + (nc * T * 3 + wave + kWaves * u) * 256, lane)[0];
      tb[u] = Wb >= 0 ? ldw4n(wbs, 2 * Wb + (nc * T * 3 + wave + kWaves * u) * 256, lane)[0] : 0.f;
    }
  }
#pragma unroll 1
  for (int cc = 0; cc < KT; cc += 3) {
    step(S0, cc);
    if (cc + 1 < KT) step(S1, cc + 1);
    if (cc + 2 < KT) step(S2, cc + 2);
  }
  // the next GEMM's first chunk travels while the K tail and the epilogue run
  if (nextW >= 0) node_prefetch_s_n<HP, NTW>(*pf, wbs, nextW, wave, lane);
  if (tail) {  // one fp32 k-step per tile and source: inputs 16 (T - 1) + g on lane group g
    const int n = c < N ? c : N - 1;
    const float xa = sXa[n * LD + 16 * (T - 1) + g];
    const float xb = Wb >= 0 ? sXb[n * LD + 16 * (T - 1) + g] : 0.f;
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      acc[u] = mfma1(ta[u], xa, acc[u]);
      if (Wb >= 0) acc[u] = mfma1(tb[u], xb, acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int t = wave + kWaves * u;
    f4 y = acc[u];
    if (c < N) {
      float* dst = sY + c * LD + 16 * t + 4 * g;
      if (gPre != nullptr) __builtin_nontemporal_store(y, (f4*)(gPre + c * HP + 16 * t + 4 * g));  // stash: write once, read once
      if (EPI == EPI_SILU) y = silu4(y);
      if (EPI == EPI_RESIDUAL_MASK) {
        const f4 r = *(const f4*)(sRes + c * LD + 16 * t + 4 * g);
        y = (r + y) * sMask[c];
      }
      if (EPI == EPI_MUL_DSILU) {  // y * silu'(pre-activation stored in sRes); in place is safe
        const f4 r = *(const f4*)(sRes + c * LD + 16 * t + 4 * g);
        y = (f4){y[0] * dsilu_f(r[0]), y[1] * dsilu_f(r[1]), y[2] * dsilu_f(r[2]), y[3] * dsilu_f(r[3])};
      }
      if (EPI == EPI_ACCUM) y = *(const f4*)(sRes + c * LD + 16 * t + 4 * g) + y;
      *(f4*)dst = y;
    }
  }
}

// One node GEMM of the workgroup (all waves call it).  sXa / sXb: the fp32 input rows (LDS [N][HP+4]); sBa / sBb: where their
// split copies live (ns_split_floats(HP) floats each); split_a / split_b: make the copy now (false: an earlier call of this
// phase already did -- P and Q share h, dh's two transposed GEMMs share dnpre).  The barrier between the copy and its readers is
// inside; the caller guarantees that nobody still reads the split buffers when the call starts (a barrier since their last use).
template <int HP, int EPI>
__device__ __forceinline__ void node_gemm_s(const WBuf& wbs, int Wa, const float* sXa, float* sBa, bool split_a, int Wb, const float* sXb,
                                            float* sBb, bool split_b, const float* sBias, float* sY, const float* sRes,
                                            const float* sMask, int N, int tid, bool ktail, NodePFS<HP>* pf, int nextW = -1,
                                            float* gPre = nullptr) {
  constexpr int T = HP / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nc = ns_chunks<HP>(ktail);
  const int KT = Wb >= 0 ? 2 * nc : nc;
  const bool own1 = wave < T, own2 = wave + kWaves < T;
  NodeASet S0, S1, S2;
  if (own1) {  // chunks 1 and 2 leave before the split: they are in flight while it runs
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p) S0.p[u][p] = pf->a[u][p];
    ns_load<HP>(S1, wbs, Wa, Wb, nc, KT, 1, wave, lane, own2);
    ns_load<HP>(S2, wbs, Wa, Wb, nc, KT, 2, wave, lane, own2);
  }
  if (split_a) node_split_rows<HP>(sBa, sXa, N, ktail, tid);
  if (split_b && Wb >= 0) node_split_rows<HP>(sBb, sXb, N, ktail, tid);
  if (split_a || split_b) lds_barrier();
  if (own2)
    node_gemm_s_body<HP, EPI, 2>(wbs, Wa, sBa, sXa, Wb, sBb, sXb, sBias, sY, sRes, sMask, N, wave, lane, ktail, S0, S1, S2, pf, nextW, gPre);
  else if (own1)
    node_gemm_s_body<HP, EPI, 1>(wbs, Wa, sBa, sXa, Wb, sBb, sXb, sBias, sY, sRes, sMask, N, wave, lane, ktail, S0, S1, S2, pf, nextW, gPre);
}

}  // namespace w8
}  // namespace gaudi
