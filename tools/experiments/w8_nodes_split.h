// w8_nodes_split.h -- EXPERIMENT (round 4, measured and rejected; tools/node_gemm4_microbench.hip is its only user): node-level
// GEMMs of the 8-wave kernels on the bf16 matrix pipe with fp32-equivalent accuracy ("NG4").
// Result (profiles/r04a_node_gemm4_microbench.txt, 256 workgroups): 8 870 cycles per H = 192 matrix against 6 358 for the fp32
// form of w8_common.h, 10 702 against 8 388 at H = 196 -> 208 (N = 11); 12 886 / 11 846 and 15 170 / 14 553 at N = 22.  Splitting
// a weight tile pair in registers costs 44 vector instructions of 4 issue cycles each for 6 matrix instructions of 16, a
// buffer_load_dwordx4 about 60 issue cycles, and on one SIMD all of them ADD: the matrix time saved (2.67x) is less than the
// vector time spent.  Numerics are at the fp32 instruction's level (same file).  DESIGN.md section 8.
//
// Y[n][o] = epi( sum_k Wa[o][k] Xa[n][k] (+ sum_k Wb[o][k] Xb[n][k]) + bias[o] ) for the N <= 32 nodes of a workgroup: the
// P / Q / node-MLP GEMMs of every layer and their transposes in the reverse pass (edm/egnn/egnn_new.py:59-73,
// edm/egnn_predictor/gcl.py:240-250) -- 183 matrices per guided step, each streamed from L2 once per workgroup.
//
// The fp32 form (w8_common.h: node_gemm, v_mfma_f32_16x16x4_f32) spends 53 % of a guided step's matrix-pipe time on 20 % of
// its useful FLOPs: the fp32 matrix instruction runs at the vector rate.  Here BOTH operands are split exactly into three
// bf16 pieces IN REGISTERS (w8_split.h: split8; x = h + m + l, six piece products, fp32 accumulate) and the products go
// through v_mfma_f32_16x16x32_bf16:
//   * the weights stay fp32 in memory (the same lane-linear tiles, 4 bytes per weight): pre-split images would stream 1.5x the
//     bytes through an L2 -> CU path that is the co-limit of these GEMMs, so a weight tile pair
//     (16 outputs x 32 inputs) is split after it has landed in registers -- 44 vector instructions for 6 matrix instructions
//     per node-column tile;
//   * the work is done by FOUR waves, one per SIMD (waves 0-3; tiles t = wave, wave + 4, wave + 8): alone on its SIMD a wave
//     issues vector instructions in the shadow of its own matrix instructions (an MFMA holds the issue port for 8 of its 16
//     cycles, MI355X_MICROARCH.md), while two waves per SIMD pay 4.3 cycles per vector instruction on top of 16.5 per MFMA
//     (profiles/r03c_coissue_microbench.txt) and would split every activation chunk twice as often.  Waves 4-7 go straight to
//     the barrier that follows every node GEMM -- except for tiles 12-15 of the widest matrices (one per wave 4-7), so that no
//     wave holds more than three tiles' weights in flight (the register budget of the phase functions);
//   * the 4-valid-row tail tile of H % 16 == 4 widths keeps its v_mfma_f32_4x4x1_16B_f32 form (w8_common.h); at the default
//     predictor width (196 -> 13 tiles) it is tile 12 = wave 4's only work, so waves 0-3 carry three full tiles each.
// Accumulation order per output element: K chunks of 32 in order, six piece products smallest first -- independent of the
// number of node columns, so a molecule's result does not depend on what shares its workgroup (packed launches).
#pragma once
#include "w8_split.h"

namespace gaudi {
namespace w8 {

constexpr int kNodeWaves = 4;     // waves 0-3 carry tiles 0-11 (t -> wave t & 3); tiles 12-15 go to waves 4-7, one each
constexpr int kNodeMaxTiles = 3;  // output tiles per wave (HP <= 256)

template <int HP>
struct NodePF4 {
  f4 a0[kNodeMaxTiles], a1[kNodeMaxTiles];  // first two 16-input chunks of the wave's tiles
};

// tiles of wave w: t = ng_tile(w, u), u < ng_ntw(T, w)
__host__ __device__ constexpr int ng_ntw(int T, int w) {
  return w < kNodeWaves ? ((T < 12 ? T : 12) - w + kNodeWaves - 1) / kNodeWaves : (w + 8 < T ? 1 : 0);
}
__host__ __device__ constexpr int ng_tile(int w, int u) { return w < kNodeWaves ? w + kNodeWaves * u : w + 8; }
__host__ __device__ constexpr int ng_wave_of(int t) { return t < 12 ? t % kNodeWaves : t - 8; }
// the wave whose LAST tile is the tail tile of a width that has one
template <int HP>
__device__ __forceinline__ bool ng_owns_tail(bool tail_width, int wave) {
  constexpr int T = HP / 16;
  constexpr bool kHas = GAUDI_NODE_TAIL44 && (HP == 208 || HP == 48);
  return kHas && tail_width && wave == ng_wave_of(T - 1);
}

template <int HP, int NTW, bool TAIL>
__device__ __forceinline__ void node_prefetch4_n(NodePF4<HP>& pf, const WBuf& wb, int W, int wave, int lane) {
  constexpr int T = HP / 16;
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int toff = ng_tile(wave, u) * 256;
    const int ln = TAIL && u == NTW - 1 ? tail_lane(lane) : lane;
    pf.a0[u] = ldw4n(wb, W + toff, ln);
    pf.a1[u] = ldw4n(wb, W + (T > 1 ? T : 0) * 256 + toff, ln);
  }
}
// wave-uniform dispatch on the wave's tile count (only the counts this width can produce are instantiated)
template <int HP, class F>
__device__ __forceinline__ void ng_dispatch(int wave, bool tail_w, F f) {
  constexpr int T = HP / 16;
  const int ntw = ng_ntw(T, wave);
  if (ntw == 0) return;
  if (ng_owns_tail<HP>(tail_w, wave)) {
    f(std::integral_constant<int, ng_ntw(T, ng_wave_of(T - 1))>{}, std::integral_constant<bool, true>{});
    return;
  }
  if constexpr (ng_ntw(T, 0) >= 3) {
    if (ntw == 3) { f(std::integral_constant<int, 3>{}, std::integral_constant<bool, false>{}); return; }
  }
  if constexpr (ng_ntw(T, 0) >= 2 && ng_ntw(T, 3) <= 2) {
    if (ntw == 2) { f(std::integral_constant<int, 2>{}, std::integral_constant<bool, false>{}); return; }
  }
  if constexpr (ng_ntw(T, 3) <= 1) {
    if (ntw == 1) { f(std::integral_constant<int, 1>{}, std::integral_constant<bool, false>{}); return; }
  }
}
template <int HP>
__device__ __forceinline__ void node_prefetch4(NodePF4<HP>& pf, const WBuf& wb, int W, int wave, int lane, bool tail_w) {
  ng_dispatch<HP>(wave, tail_w, [&](auto ntw_tag, auto tail_tag) {
    node_prefetch4_n<HP, decltype(ntw_tag)::value, decltype(tail_tag)::value>(pf, wb, W, wave, lane);
  });
}

// six piece products of a (weights: A operand) x b (activations: B operand), smallest first (w8_split.h: rings_mfma_act)
__device__ __forceinline__ f4 mfma6(const B3& a, const B3& b, f4 c) {
  c = mfma_bf(a.l, b.h, c);
  c = mfma_bf(a.h, b.l, c);
  c = mfma_bf(a.m, b.m, c);
  c = mfma_bf(a.m, b.h, c);
  c = mfma_bf(a.h, b.m, c);
  c = mfma_bf(a.h, b.h, c);
  return c;
}

template <int HP, int EPI, bool PRE, int NT, int NTW, bool TAIL>
__device__ __forceinline__ void node_gemm4_body(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb,
                                                const float* sBias, float* sY, const float* sRes, const float* sMask, int N,
                                                int wave, int lane, NodePF4<HP>* pf, int nextW, float* gPre) {
  constexpr int T = HP / 16;
  constexpr int LD = HP + 4;
  const int c = lane & 15, g = lane >> 4;
  const int n_tiles = (N + 15) >> 4;
  int toff[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) toff[u] = ng_tile(wave, u) * 256;
  auto wlane = [&](int u) { return TAIL && u == NTW - 1 ? tail_lane_fresh(lane) : lane; };
  const int KT = Wb >= 0 ? 2 * T : T;  // 16-input chunks; two sources run as ONE K loop so the load pipeline never restarts
  auto chunk = [&](int cc) {           // float offset of chunk cc (clamped past the end: surplus loads are unused)
    const int k = cc < KT ? cc : KT - 1;
    return k < T ? Wa + k * (T * 256) : Wb + (k - T) * (T * 256);
  };
  for (int nt0 = 0; nt0 < n_tiles; nt0 += NT) {
    const float* xa[NT];
    const float* xb[NT];
    int node[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      node[j] = (nt0 + j) * 16 + c;
      const int nclamp = node[j] < N ? node[j] : N - 1;
      xa[j] = sXa + nclamp * LD + 4 * g;
      xb[j] = Wb >= 0 ? sXb + nclamp * LD + 4 * g - 16 * T : xa[j];  // indexed by the global chunk number
    }
    auto xin = [&](int j, int cc) { return *(const f4*)((cc < T ? xa[j] : xb[j]) + 16 * cc); };
    f4 acc[NT][NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      // the tail tile's lane groups hold partial sums of the SAME outputs: its bias is added after they are folded
      const f4 b = sBias != nullptr && !(TAIL && u == NTW - 1) ? *(const f4*)(sBias + (toff[u] >> 4) + 4 * g) : splat(0.f);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j][u] = b;
    }
    f4 a0[NTW], a1[NTW], b0[NTW], b1[NTW];  // ping-pong sets of one 32-input chunk each (= two 16-input tiles per owned tile)
    if (PRE && nt0 == 0) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) { a0[u] = pf->a0[u]; a1[u] = pf->a1[u]; }
    } else {
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        a0[u] = ldw4n(wb, chunk(0) + toff[u], wlane(u));
        a1[u] = ldw4n(wb, chunk(1) + toff[u], wlane(u));
      }
    }
    // one 32-input chunk: inputs 16 cc .. +15 (wE, xE) and 16 (cc + 1) .. +15 (wO, xO; ODD = false: absent)
    auto mm = [&](auto odd_tag, const f4 (&wE)[NTW], const f4 (&xE)[NT], const f4 (&wO)[NTW], const f4 (&xO)[NT]) {
      constexpr bool ODD = decltype(odd_tag)::value;
      constexpr int NS = TAIL ? NTW - 1 : NTW;  // tiles on split operands (the tail tile, if any, is the wave's last)
      B3 xs[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) xs[j] = split8(xE[j], ODD ? xO[j] : splat(0.f));
      if constexpr (TAIL) {  // the 4-valid-row tile first: fp32 4x4x1 blocks (w8_common.h), no split needed
        constexpr int u = NTW - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j][u] = mfma44(wE[u][q], xE[j][q], acc[j][u]);
        if (ODD) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][u] = mfma44(wO[u][q], xO[j][q], acc[j][u]);
        }
      }
      // software pipeline over the wave's tiles: the split of tile u + 1 (44 vector instructions) is issued beside the six
      // matrix instructions per node column of tile u; one scheduling region per tile keeps hipcc from splitting all tiles
      // up front (12 more live registers per tile)
      if constexpr (NS > 0) {
        B3 ws = split8(wE[0], ODD ? wO[0] : splat(0.f));
        static_for_n<NS>([&](auto u_tag) {
          constexpr int u = decltype(u_tag)::value;
          B3 wn = ws;
          if constexpr (u + 1 < NS) wn = split8(wE[u + 1], ODD ? wO[u + 1] : splat(0.f));
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j][u] = mfma6(ws, xs[j], acc[j][u]);
          __builtin_amdgcn_sched_barrier(0);
          ws = wn;
        });
      }
    };
    using Odd = std::integral_constant<bool, true>;
    using Even = std::integral_constant<bool, false>;
    const int main_end = KT / 4 * 4;
#pragma unroll 1
    for (int cc = 0; cc < main_end; cc += 4) {
      f4 x0[NT], x1[NT], x2[NT], x3[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        x0[j] = xin(j, cc);
        x1[j] = xin(j, cc + 1);
        x2[j] = xin(j, cc + 2);
        x3[j] = xin(j, cc + 3);
      }
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        b0[u] = ldw4n(wb, chunk(cc + 2) + toff[u], wlane(u));
        b1[u] = ldw4n(wb, chunk(cc + 3) + toff[u], wlane(u));
      }
      __builtin_amdgcn_sched_barrier(0);  // LDS reads + set B loads | chunk on set A | set A loads | chunk on set B
      mm(Odd{}, a0, x0, a1, x1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        a0[u] = ldw4n(wb, chunk(cc + 4) + toff[u], wlane(u));
        a1[u] = ldw4n(wb, chunk(cc + 5) + toff[u], wlane(u));
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(Odd{}, b0, x2, b1, x3);
      __builtin_amdgcn_sched_barrier(0);
    }
    // remainder: KT % 4 16-input chunks (0..3), the first two already in a0 / a1
    const int rem = KT - main_end;
    if (rem >= 3) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) b0[u] = ldw4n(wb, chunk(main_end + 2) + toff[u], wlane(u));
    }
    if (rem >= 2) {
      f4 x0[NT], x1[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) { x0[j] = xin(j, main_end); x1[j] = xin(j, main_end + 1); }
      mm(Odd{}, a0, x0, a1, x1);
    } else if (rem == 1) {
      f4 x0[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) x0[j] = xin(j, main_end);
      mm(Even{}, a0, x0, a0, x0);
    }
    // software pipelining ACROSS calls: the next node GEMM's first tiles travel while this one drains
    if (nextW >= 0 && nt0 + NT >= n_tiles) node_prefetch4_n<HP, NTW, TAIL>(*pf, wb, nextW, wave, lane);
    if (rem >= 3) {
      f4 x2[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) x2[j] = xin(j, main_end + 2);
      mm(Even{}, b0, x2, b0, x2);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        const int t = ng_tile(wave, u);
        const int nd = node[j];
        f4 y = acc[j][u];
        if (TAIL && u == NTW - 1) {  // fold the four k partial sums (all lanes take part), then bias; padding rows = 0
          y = (f4){reduce_groups(y[0]), reduce_groups(y[1]), reduce_groups(y[2]), reduce_groups(y[3])};
          if (sBias != nullptr) y = y + *(const f4*)(sBias + 16 * t);
        }
        if (nd < N) {
          float* dst = sY + nd * LD + 16 * t + 4 * g;
          if (TAIL && u == NTW - 1 && g > 0) {
            if (gPre != nullptr) __builtin_nontemporal_store(splat(0.f), (f4*)(gPre + nd * HP + 16 * t + 4 * g));
            *(f4*)dst = splat(0.f);
            continue;
          }
          if (gPre != nullptr) __builtin_nontemporal_store(y, (f4*)(gPre + nd * HP + 16 * t + 4 * g));  // stash: write once, read once
          if (EPI == EPI_SILU) y = silu4(y);
          if (EPI == EPI_RESIDUAL_MASK) {
            const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g);
            y = (r + y) * sMask[nd];
          }
          if (EPI == EPI_MUL_DSILU) {  // y * silu'(pre-activation stored in sRes); in place is safe
            const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g);
            y = (f4){y[0] * dsilu_f(r[0]), y[1] * dsilu_f(r[1]), y[2] * dsilu_f(r[2]), y[3] * dsilu_f(r[3])};
          }
          if (EPI == EPI_ACCUM) y = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g) + y;
          *(f4*)dst = y;
        }
      }
  }
}

template <int HP, int EPI, bool PRE, int NTW, bool TAIL>
__device__ __forceinline__ void node_gemm4_cols(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb,
                                                const float* sBias, float* sY, const float* sRes, const float* sMask, int N,
                                                int wave, int lane, NodePF4<HP>* pf, int nextW, float* gPre) {
  if (N <= 16) node_gemm4_body<HP, EPI, PRE, 1, NTW, TAIL>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
  else node_gemm4_body<HP, EPI, PRE, 2, NTW, TAIL>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
}

// same interface as w8_common.h: node_gemm (pf: NodePF4).  Waves without a tile return at once: the caller's barrier collects them.
template <int HP, int EPI, bool PRE = false>
__device__ __forceinline__ void node_gemm4(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb,
                                           const float* sBias /* LDS [HP] or null */, float* sY, const float* sRes,
                                           const float* sMask, int N, int wave, int lane, bool tail_w,
                                           NodePF4<HP>* pf = nullptr, int nextW = -1, float* gPre = nullptr) {
  ng_dispatch<HP>(wave, tail_w, [&](auto ntw_tag, auto tail_tag) {
    node_gemm4_cols<HP, EPI, PRE, decltype(ntw_tag)::value, decltype(tail_tag)::value>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave,
                                                                                       lane, pf, nextW, gPre);
  });
}

}  // namespace w8
}  // namespace gaudi
