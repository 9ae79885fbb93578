// w8_common.h -- building blocks of the EIGHT-wave sampler kernels (two wavefronts per SIMD, <= 256 registers each).
//
// Same contraction orientation as device_common.h (weights = MFMA A operand, graph nodes / edges on the 16 columns), but:
//   * one molecule is served by 8 waves; an edge-level GEMM gives every wave ONE 16-edge tile (tile tau -> wave tau & 7,
//     round tau >> 3), so each SIMD hosts two waves whose MFMAs, LDS round trips and epilogues cover each other;
//   * the weight stream of an edge-level GEMM is fetched from L2 ONCE per workgroup: the T tiles of a K chunk are dealt to
//     the 8 waves, staged through registers (issued a whole K chunk early, written after the barrier) into a two-slot LDS
//     ring, and every wave reads its A fragments from LDS (lane-linear 1 KiB tiles: conflict-free ds_read_b128);
//   * the edge -> node sum is a segmented scan inside the 16-lane DPP rows of the accumulator registers (run boundaries
//     come from the host's edge list, sorted by receiving node): no LDS transposition scratch, no atomics, fixed order;
//   * node-level GEMMs deal their output-feature tiles to 8 waves (weights straight from L2, each tile read by one wave).
// Weight tiles are packed "lane-linear": float4 index L of a 16x16 tile holds W[row = L & 15][k = 4 * (L >> 4) .. +3].
#pragma once
#include <type_traits>

#include "device_common.h"

namespace gaudi {
namespace w8 {

constexpr int kWaves = 8;
constexpr int kThreads = 512;

// ---------------------------------------------------------------------------------------------
// DPP row shifts (rows = 16 lanes = the 16 edge columns of a tile; lanes shifted in from outside the row read 0)
// ---------------------------------------------------------------------------------------------
template <int SH>
__device__ __forceinline__ float row_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SH, 0xF, 0xF, true));
}

template <int N, class F>
__device__ __forceinline__ void static_for_n(F f) {
  if constexpr (N > 0) {
    static_for_n<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Run bookkeeping of one 16-edge tile, per lane (column c = lane & 15): the edge list is sorted by receiving node, so the
// edges of a node form a run of consecutive columns.  m1..m8 = 1 where the column 1/2/4/8 to the left belongs to the same
// run (Hillis-Steele segmented scan); after seg_scan the LAST column of every run holds the run's sum.
struct RunMask {
  float m1, m2, m4, m8;
  __device__ __forceinline__ void set(int c, int run_start) {
    m1 = c - 1 >= run_start ? 1.f : 0.f;
    m2 = c - 2 >= run_start ? 1.f : 0.f;
    m4 = c - 4 >= run_start ? 1.f : 0.f;
    m8 = c - 8 >= run_start ? 1.f : 0.f;
  }
};
__device__ __forceinline__ float seg_scan1(float x, const RunMask& m) {
  x = fmaf(row_shr<1>(x), m.m1, x);
  x = fmaf(row_shr<2>(x), m.m2, x);
  x = fmaf(row_shr<4>(x), m.m4, x);
  x = fmaf(row_shr<8>(x), m.m8, x);
  return x;
}
__device__ __forceinline__ f4 seg_scan(f4 v, const RunMask& m) {
  return (f4){seg_scan1(v[0], m), seg_scan1(v[1], m), seg_scan1(v[2], m), seg_scan1(v[3], m)};
}

// edge word of a slot: i | j << 8 | run_start << 16 (column inside the tile) | run_end << 20 | part << 21
__device__ __forceinline__ int ew_i(uint32_t e) { return e & 255; }
__device__ __forceinline__ int ew_j(uint32_t e) { return (e >> 8) & 255; }
__device__ __forceinline__ int ew_run_start(uint32_t e) { return (e >> 16) & 15; }
__device__ __forceinline__ bool ew_run_end(uint32_t e) { return (e >> 20) & 1; }
__device__ __forceinline__ int ew_part(uint32_t e) { return (e >> 21) & 1; }

// ---------------------------------------------------------------------------------------------
// Weight ring of the edge-level GEMMs.  A matrix is T x T tiles, k-major ([k/16][o/16] like every packed matrix); group g
// = the T tiles of K chunk g, dealt to the 8 waves (tile t -> wave t & 7).  Two LDS slots + one group in registers.
// Invariant at the barrier that opens a trip: slot(par) holds the group the trip consumes, nobody reads slot(par ^ 1) any
// more, and `st` holds the following group (loads issued one trip ago).  In the MIDDLE of its MFMA block every wave stores
// `st` to slot(par ^ 1) and re-issues the loads of the group after that into the same registers.
//   * ONE unconditional program point defines and one uses the in-flight registers per trip: with the loads defined on
//     two control paths hipcc merges them through register copies, i.e. a vmcnt(0) wait per trip; no branches around the
//     loads either (out-of-range lane offsets instead: such a load returns 0 and fetches nothing);
//   * mid-block, because that is where a wave's stalls (the CU's vector-memory pipe accepts ~31 B/clk, all eight waves
//     issue their 1 KiB loads together) are covered by its SIMD partner's MFMAs, and every wave still ENDS its trip on
//     MFMAs, so nobody holds the next barrier back.
// ---------------------------------------------------------------------------------------------
template <int HP>
struct Ring {
  static constexpr int T = HP / 16;
  static constexpr int UT = (T + kWaves - 1) / kWaves;
  static constexpr int kSlotFloats = T * 256;
  float* base;  // LDS [2][T * 256]
  int par;
  bool ktail;   // last K chunk = one k-step (device_common.h: edge_u_tail)
  f4 st[UT];
  __device__ __forceinline__ float* slot(int p) const { return base + p * kSlotFloats; }
};

constexpr int kOOBLane = 0x0FFFFFFF;  // lane offset beyond any descriptor's range

template <int HP>
__device__ __forceinline__ void ring_issue(Ring<HP>& r, const WBuf& wb, int group_off, bool have, int wave, int lane) {
  group_off = __builtin_amdgcn_readfirstlane(group_off);
#pragma unroll
  for (int u = 0; u < Ring<HP>::UT; ++u) {
    const int t = wave + kWaves * u;
    r.st[u] = ldw4(wb, group_off + (t < Ring<HP>::T ? t : 0) * 256, (have && t < Ring<HP>::T) ? lane : kOOBLane);
  }
}
template <int HP>
__device__ __forceinline__ void ring_commit(const Ring<HP>& r, float* slot, int wave, int lane) {
#pragma unroll
  for (int u = 0; u < Ring<HP>::UT; ++u) {
    const int t = wave + kWaves * u;
    if (t < Ring<HP>::T) *(f4*)(slot + t * 256 + lane * 4) = r.st[u];
  }
}
// Open a chain of edge GEMMs with matrix W.  PRE: no wave still reads slot(par) (a barrier lies in between).
template <int HP>
__device__ __forceinline__ void ring_start(Ring<HP>& r, const WBuf& wb, int W, int wave, int lane) {
  ring_issue(r, wb, W, true, wave, lane);
  ring_commit(r, r.slot(r.par), wave, lane);
  ring_issue(r, wb, W + Ring<HP>::kSlotFloats, true, wave, lane);
}
// Staging step of trip cc of a GEMM with matrix W (next matrix of the chain: nextW, -1 = none)
template <int HP>
__device__ __forceinline__ void ring_stage(Ring<HP>& r, const WBuf& wb, int W, int nextW, int cc, int wave, int lane) {
  constexpr int T = HP / 16;
  ring_commit(r, r.slot(r.par ^ 1), wave, lane);  // at the end of a chain this parks don't-care data in the free slot
  const int g2 = cc + 2;
  const int off = g2 < T ? W + g2 * Ring<HP>::kSlotFloats : nextW + (g2 - T) * Ring<HP>::kSlotFloats;
  ring_issue(r, wb, off, g2 < T || nextW >= 0, wave, lane);
}

// The T tiles of the current slot against one input chunk: pairs of output tiles, k-step outermost (a dependent accumulate
// needs 40 cycles, issue is every 32).  A fragments are read one pair ahead (explicit double buffer + fences: left alone,
// hipcc sinks every ds_read next to its MFMAs and exposes the LDS latency once per pair).  `mid` runs between the two
// halves of the block for every wave; a wave without a tile in this round (`active` false) skips only the MFMAs.
// The last accumulator tile of a chained GEMM as the B operand of a K-tail chunk: lane (column c, group g) takes element g of
// group 0's lane c (features 16(T-1) .. +3 live there), element 0.
// (__shfl, not a hand-written ds_bpermute + integer select: hipcc folded that form to "element 0 for every group" --
// /tmp-probe verified on the GPU; the __shfl form moves the right elements)
__device__ __forceinline__ f4 tail_to_b(f4 v, int c, int g) {
  const float t0 = __shfl(v[0], c, 64), t1 = __shfl(v[1], c, 64), t2 = __shfl(v[2], c, 64), t3 = __shfl(v[3], c, 64);
  return (f4){g == 0 ? t0 : g == 1 ? t1 : g == 2 ? t2 : t3, 0.f, 0.f, 0.f};
}

// NQ = k-steps of this chunk: 4, or 1 for a K-tail chunk (compile-time: a run-time count splits the MFMA block into
// basic blocks and costs far more than the tail saves)
template <int HP, int NQ, class MID>
__device__ __forceinline__ void ring_mfma(f4 (&acc)[HP / 16], const float* slot_lane, const f4 bin, bool active, MID mid) {
  constexpr int T = HP / 16;
  constexpr int NP = (T + 1) / 2;  // pairs of output tiles
  f4 a[2][2];
  a[0][0] = *(const f4*)(slot_lane);
  a[0][1] = T > 1 ? *(const f4*)(slot_lane + 256) : a[0][0];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int t0 = 2 * p, cur = p & 1;
    if (p == NP / 2) {
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (p + 1 < NP) {
      a[cur ^ 1][0] = *(const f4*)(slot_lane + (t0 + 2) * 256);
      a[cur ^ 1][1] = t0 + 3 < T ? *(const f4*)(slot_lane + (t0 + 3) * 256) : a[cur ^ 1][0];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        acc[t0] = mfma1(a[cur][0][q], bin[q], acc[t0]);
        if (t0 + 1 < T) acc[t0 + 1] = mfma1(a[cur][1][q], bin[q], acc[t0 + 1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[t] (features 16t+4g+q of the lane's edge column) = b2 + W2 . silu(u),  u = P[i] + Q[j] + cr r + cd d0  (b1 inside P):
// Linear(2H+2 -> H) of [h_i | h_j | r | d0] factorised per node (egnn_new.py:42-47,119-129; egnn_predictor/gcl.py:225-231).
// pp / qq: the lane's P and Q rows (+ 4g).  One trip = barrier | MFMAs on group cc with the ring staging in the middle.
// The two waves of a SIMD generate their input chunk at different points (waves 4-7 right after the barrier, waves 0-3 in the
// middle of the block, for the NEXT trip): measured 3-4 % better than any common placement (tools/edge_gemm_microbench.hip).
// What the partner wave hides is LATENCY (LDS, VMEM issue, barrier skew); vector-ALU work is NOT hidden behind fp32 MFMAs --
// v_mfma_f32_16x16x4_f32 runs at the vector fp32 rate and SQ_VALU_MFMA_COEXEC_CYCLES stays 0 -- so a trip costs its MFMA
// time plus its VALU time (3832 cycles per trip of 2 x 48 MFMAs against 3072; 3576 without the input generation).
template <int HP>
__device__ __forceinline__ void edge_gemm_pq(f4 (&acc)[HP / 16], Ring<HP>& ring, const WBuf& wb, int W, int nextW,
                                             const float* sB2, const float* sCr, const float* sCd, const float* pp,
                                             const float* qq, float r, float d0, bool active, int wave, int lane STAMP_DECL) {
  constexpr int T = HP / 16;
  const int g = lane >> 4;
  const bool late = wave >= kWaves / 2;
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = *(const f4*)(sB2 + 16 * t + 4 * g);
  const bool ktail = ring.ktail;
  auto gen = [&](int cc) { return silu4v(edge_u(pp, qq, sCr, sCd, g, cc, r, d0)); };  // silu(u) of input chunk cc, B layout
  auto gen_last = [&] {  // ... of the last chunk: one k-step when the matrix carries a K tail
    return ktail ? silu4v(edge_u_tail(pp, qq, sCr, sCd, g, T, r, d0)) : gen(T - 1);
  };
  f4 bin = T > 1 ? gen(0) : gen_last(), nb = bin;
  // one trip; NQ k-steps; `last_next`: the chunk generated for the next trip is the last one
  auto trip = [&](int cc, auto nq_tag, bool cur_last, bool next_last) {
    constexpr int NQ = decltype(nq_tag)::value;
    __syncthreads();
    if (late && cc > 0) bin = cur_last ? gen_last() : gen(cc);
    ring_mfma<HP, NQ>(acc, ring.slot(ring.par) + lane * 4, bin, active, [&] {
      ring_stage<HP>(ring, wb, W, nextW, cc, wave, lane);
      if (!late) nb = next_last ? gen_last() : gen(cc + 1 < T ? cc + 1 : T - 1);
    });
    if (!late) bin = nb;
    ring.par ^= 1;
  };
  using Q4 = std::integral_constant<int, 4>;
  using Q1 = std::integral_constant<int, 1>;
#pragma unroll 1
  for (int cc = 0; cc < T - 2; ++cc) trip(cc, Q4{}, false, false);
  if (T > 1) trip(T - 2, Q4{}, false, true);  // generates the last chunk for waves 0-3
  if (ktail) trip(T - 1, Q1{}, true, true);
  else trip(T - 1, Q4{}, true, true);
}

// Chained edge GEMM, input already in registers in C/B layout: out = bias + rowinit + W . in
template <int HP>
__device__ __forceinline__ void edge_gemm_regs(f4 (&out)[HP / 16], const f4 (&in)[HP / 16], Ring<HP>& ring, const WBuf& wb,
                                               int W, int nextW, const float* sBias, const float* rowinit, bool active,
                                               int wave, int lane) {
  constexpr int T = HP / 16;
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f4 b = sBias != nullptr ? *(const f4*)(sBias + 16 * t + 4 * g) : splat(0.f);
    if (rowinit != nullptr) b = b + *(const f4*)(rowinit + 16 * t + 4 * g);
    out[t] = b;
  }
  const int c = lane & 15;
#pragma unroll
  for (int cc = 0; cc < T; ++cc) {
    __syncthreads();
    auto stage = [&] { ring_stage<HP>(ring, wb, W, nextW, cc, wave, lane); };
    if (cc == T - 1 && ring.ktail)
      ring_mfma<HP, 1>(out, ring.slot(ring.par) + lane * 4, tail_to_b(in[cc], c, g), active, stage);
    else
      ring_mfma<HP, 4>(out, ring.slot(ring.par) + lane * 4, in[cc], active, stage);
    ring.par ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Node-level GEMM on 8 waves:  Y[n][o] = epi( sum_k Wa[o][k] Xa[n][k] (+ sum_k Wb[o][k] Xb[n][k]) + bias[o] )
// Output-feature tile t belongs to wave t & 7 (tiles t = wave, wave + 8): a wave owns NTW = 0, 1 or 2 tiles and the body
// is instantiated per NTW (a wave-uniform branch picks it), so no wave issues MFMAs for tiles that do not exist -- with two
// waves per SIMD a dummy tile would steal matrix time from the partner.
// ---------------------------------------------------------------------------------------------
template <int HP>
struct NodePF {
  f4 a0[2], a1[2];  // first two K chunks of the wave's (up to) two tiles
};

// The "tail tile" of a hidden size with H % 16 == 4 (196 -> 208, 36 -> 48): output tile T-1 holds 4 valid rows.  As a
// 16x16x4 tile it costs as much matrix time as a full one and puts a fourth tile on one SIMD (13 tiles on 8 waves) while
// the others carry three.  It is computed with v_mfma_f32_4x4x1_16B_f32 instead: 16 independent 4x4 outer products per
// instruction, block (kk, cg) = lane >> 2 holding rows 0-3 x node columns 4 cg .. +3 for input 4 kk + q of the chunk --
// four inputs per instruction like the 16x16x4 form, at a quarter of its cycles.  Lane (kk, cg, i) needs
// W[16 (T-1) + i][16 m + 4 kk + q], q = 0..3: float4 index kk * 16 + i of the SAME lane-linear tile (so no second weight
// layout), and the activation of node 4 cg + (lane & 3) = lane & 15 at inputs 16 m + 4 kk + q: exactly the float4 the
// 16x16x4 form reads.  The four kk partial sums of an output sit in the four lane groups and are added at the end
// (reduce_groups): accumulator register r of lane (c, any group) = feature 16 (T-1) + r of node c, the C layout of group 0.
// Features 16 (T-1) + 4 .. +15 are padding and are written as zeros.
__device__ __forceinline__ int tail_lane(int lane) { return (lane & 0x30) | (lane & 3); }
// ... recomputed at every load (two vector instructions) instead of living in a register beside `lane` through the K loop:
// the node GEMMs run at the register limit and a spilled value is reloaded through the same in-order vmcnt queue as the
// weight prefetch
__device__ __forceinline__ int tail_lane_fresh(int lane) {
  asm volatile("" : "+v"(lane));
  return tail_lane(lane);
}
__device__ __forceinline__ f4 mfma44(float w, float b, f4 acc) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(w, b, acc, 0, 0, 0);
}
#ifndef GAUDI_NODE_TAIL44
#define GAUDI_NODE_TAIL44 1
#endif
// the wave (if any) whose LAST tile is the tail tile, for widths that have one
template <int HP>
__device__ __forceinline__ bool owns_tail(bool tail_width, int wave) {
  constexpr int T = HP / 16;
  // instantiated for the padded widths whose usual hidden size has a tail (196 -> 208; 36 -> 48 in the tests): every
  // extra body costs registers in all of them
  constexpr bool kHas = GAUDI_NODE_TAIL44 && (HP == 208 || HP == 48);
  return kHas && tail_width && wave == (T - 1) % kWaves;
}

template <int HP, int NTW, bool TAIL = false>
__device__ __forceinline__ void node_prefetch_n(NodePF<HP>& pf, const WBuf& wb, int W, int wave, int lane) {
  constexpr int T = HP / 16;
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int toff = (wave + kWaves * u) * 256;
    const int ln = TAIL && u == NTW - 1 ? tail_lane(lane) : lane;
    pf.a0[u] = ldw4n(wb, W + toff, ln);
    pf.a1[u] = ldw4n(wb, W + (T > 1 ? T : 0) * 256 + toff, ln);
  }
}
template <int HP>
__device__ __forceinline__ void node_prefetch(NodePF<HP>& pf, const WBuf& wb, int W, int wave, int lane, bool tail_w) {
  constexpr int T = HP / 16;
  if (owns_tail<HP>(tail_w, wave)) {
    node_prefetch_n<HP, (T - 1 >= kWaves ? 2 : 1), true>(pf, wb, W, wave, lane);  // the owner's tile count is a constant
  } else if (wave + kWaves < T) node_prefetch_n<HP, 2>(pf, wb, W, wave, lane);
  else if (wave < T) node_prefetch_n<HP, 1>(pf, wb, W, wave, lane);
}

template <int HP, int EPI, bool PRE, int NT, int NTW, bool TAIL = false>
__device__ __forceinline__ void node_gemm_body(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb,
                                               const float* sBias, float* sY, const float* sRes, const float* sMask, int N,
                                               int wave, int lane, NodePF<HP>* pf, int nextW, float* gPre) {
  constexpr int T = HP / 16;
  constexpr int LD = HP + 4;
  const int c = lane & 15, g = lane >> 4;
  const int n_tiles = (N + 15) >> 4;
  int toff[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) toff[u] = (wave + kWaves * u) * 256;
  // the lane's float4 inside weight tile u (TAIL: the last tile is read in the 4x4 block arrangement)
  auto wlane = [&](int u) { return TAIL && u == NTW - 1 ? tail_lane_fresh(lane) : lane; };
  // one matrix instruction of tile u: 16x16x4, or 4x4x1 x 16 blocks for the tail tile
  auto fma_u = [&](auto u_tag, float w, float x, f4 a) {
    constexpr int u = decltype(u_tag)::value;
    if constexpr (TAIL && u == NTW - 1) return mfma44(w, x, a);
    else return mfma1(w, x, a);
  };
  const int KT = Wb >= 0 ? 2 * T : T;  // two sources run as ONE K loop so the load pipeline never restarts
  auto chunk = [&](int cc) {  // float offset of K chunk cc (clamped past the end: surplus loads are unused)
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
    // One accumulator per (column tile, output tile); a wave that owns a single output tile of a single column tile gets a
    // second one for the odd K chunks (a dependent accumulate needs 40 cycles, issue is every 32).
    // (a wave with one output tile: even K chunks go to one accumulator, odd ones to a second -- for BOTH column-tile counts,
    // so that a node's result does not depend on how many other nodes share the workgroup: packed launches)
    constexpr bool kSplit = NTW == 1;
    constexpr int NA = kSplit ? 2 : 1;
    f4 acc[NA][NT][NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      // the tail tile's lane groups hold partial sums of the SAME outputs: its bias is added after they are folded
      const f4 b = sBias != nullptr && !(TAIL && u == NTW - 1) ? *(const f4*)(sBias + (toff[u] >> 4) + 4 * g) : splat(0.f);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[0][j][u] = b;
        if (kSplit) acc[NA - 1][j][u] = splat(0.f);
      }
    }
    f4 a0[NTW], a1[NTW], b0[NTW], b1[NTW];  // ping-pong sets of two K chunks each (roles swapped by unrolling)
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
    // two K chunks: k-step outermost over the independent accumulators
    auto mm2 = [&](const f4 (&wE)[NTW], const f4 (&xE)[NT], const f4 (&wO)[NTW], const f4 (&xO)[NT]) {
      if (kSplit) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[0][j][0] = fma_u(std::integral_constant<int, 0>{}, wE[0][q], xE[j][q], acc[0][j][0]);
            acc[NA - 1][j][0] = fma_u(std::integral_constant<int, 0>{}, wO[0][q], xO[j][q], acc[NA - 1][j][0]);
          }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            static_for_n<NTW>([&](auto u_tag) {
              constexpr int u = decltype(u_tag)::value;
              acc[0][j][u] = fma_u(u_tag, wE[u][q], xE[j][q], acc[0][j][u]);
            });
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            static_for_n<NTW>([&](auto u_tag) {
              constexpr int u = decltype(u_tag)::value;
              acc[0][j][u] = fma_u(u_tag, wO[u][q], xO[j][q], acc[0][j][u]);
            });
      }
    };
    auto mm1 = [&](const f4 (&w)[NTW], int cc) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f4 x = xin(j, cc);
          static_for_n<NTW>([&](auto u_tag) {
            constexpr int u = decltype(u_tag)::value;
            acc[0][j][u] = fma_u(u_tag, w[u][q], x[q], acc[0][j][u]);
          });
        }
    };
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
      __builtin_amdgcn_sched_barrier(0);  // LDS reads + set B loads | MFMAs on set A | set A loads | MFMAs on set B
      mm2(a0, x0, a1, x1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        a0[u] = ldw4n(wb, chunk(cc + 4) + toff[u], wlane(u));
        a1[u] = ldw4n(wb, chunk(cc + 5) + toff[u], wlane(u));
      }
      __builtin_amdgcn_sched_barrier(0);
      mm2(b0, x2, b1, x3);
      __builtin_amdgcn_sched_barrier(0);
    }
    // tail: KT % 4 chunks (0..3), the first two already in a0 / a1
    const int rem = KT - main_end;
    if (rem >= 3) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) b0[u] = ldw4n(wb, chunk(main_end + 2) + toff[u], wlane(u));
    }
    if (rem >= 1) mm1(a0, main_end);
    if (rem >= 2) mm1(a1, main_end + 1);
    // software pipelining ACROSS calls: the next node GEMM's first tiles travel while this one drains
    if (nextW >= 0 && nt0 + NT >= n_tiles) node_prefetch_n<HP, NTW, TAIL>(*pf, wb, nextW, wave, lane);
    if (rem >= 3) mm1(b0, main_end + 2);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        const int t = wave + kWaves * u;
        const int nd = node[j];
        f4 y = kSplit ? acc[0][j][u] + acc[NA - 1][j][u] : acc[0][j][u];
        if (TAIL && u == NTW - 1) {  // fold the four k partial sums (all lanes take part), then bias; padding rows = 0
          y = (f4){reduce_groups(y[0]), reduce_groups(y[1]), reduce_groups(y[2]), reduce_groups(y[3])};
          if (sBias != nullptr) y = y + *(const f4*)(sBias + 16 * t);
        }
        if (nd < N) {
          float* dst = sY + nd * LD + 16 * t + 4 * g;
          if (TAIL && u == NTW - 1 && g > 0) {
            if (gPre != nullptr) nstash_store((f4*)(gPre + nd * HP + 16 * t + 4 * g), splat(0.f));
            *(f4*)dst = splat(0.f);
            continue;
          }
          if (gPre != nullptr) nstash_store((f4*)(gPre + nd * HP + 16 * t + 4 * g), y);  // stash: write once, read once
          if (EPI == EPI_SILU) y = silu4v(y);
          if (EPI == EPI_RESIDUAL_MASK) {
            const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g);
            y = (r + y) * sMask[nd];
          }
          if (EPI == EPI_MUL_DSILU) {  // y * silu'(pre-activation stored in sRes); in place is safe
            const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g);
            y = y * dsilu4v(r);
          }
          if (EPI == EPI_ACCUM) y = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g) + y;
          *(f4*)dst = y;
        }
      }
  }
}

template <int HP, int EPI, bool PRE = false>
__device__ __forceinline__ void node_gemm(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb,
                                          const float* sBias /* LDS [HP] or null */, float* sY, const float* sRes,
                                          const float* sMask, int N, int wave, int lane,
                                          bool tail_w /* H % 16 == 4: the last output tile holds 4 valid rows (tail_lane) */,
                                          NodePF<HP>* pf = nullptr, int nextW = -1,
                                          float* gPre = nullptr /* global [N][HP]: pre-epilogue value */) {
  constexpr int T = HP / 16;
  if (owns_tail<HP>(tail_w, wave)) {
    constexpr int NTW = T - 1 >= kWaves ? 2 : 1;
    if (N <= 16)
      node_gemm_body<HP, EPI, PRE, 1, NTW, true>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
    else
      node_gemm_body<HP, EPI, PRE, 2, NTW, true>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
  } else if (wave + kWaves < T) {
    if (N <= 16)
      node_gemm_body<HP, EPI, PRE, 1, 2>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
    else
      node_gemm_body<HP, EPI, PRE, 2, 2>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
  } else if (wave < T) {
    if (N <= 16)
      node_gemm_body<HP, EPI, PRE, 1, 1>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
    else
      node_gemm_body<HP, EPI, PRE, 2, 1>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
  }
}

// node_gemm with up to MAXNT column tiles per pass.  MAXNT = 3 (the kernels with node buffers in global memory, whose molecules
// have up to 48 nodes): 33..48 nodes run as ONE pass over three column tiles instead of two passes that stream the weights
// twice.  A node's result does not depend on the column-tile count (one accumulator chain per (column tile, output tile)).
template <int HP, int EPI, bool PRE, int MAXNT>
__device__ __forceinline__ void node_gemm_n(const WBuf& wb, int Wa, const float* sXa, int Wb, const float* sXb, const float* sBias,
                                            float* sY, const float* sRes, const float* sMask, int N, int wave, int lane, bool tail_w,
                                            NodePF<HP>* pf = nullptr, int nextW = -1, float* gPre = nullptr) {
  constexpr int T = HP / 16;
  if constexpr (MAXNT < 3) {
    node_gemm<HP, EPI, PRE>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, tail_w, pf, nextW, gPre);
  } else {
    if (N <= 32 || N > 48) {
      node_gemm<HP, EPI, PRE>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, tail_w, pf, nextW, gPre);
    } else if (owns_tail<HP>(tail_w, wave)) {
      node_gemm_body<HP, EPI, PRE, 3, (T - 1 >= kWaves ? 2 : 1), true>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf,
                                                                       nextW, gPre);
    } else if (wave + kWaves < T) {
      node_gemm_body<HP, EPI, PRE, 3, 2>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
    } else if (wave < T) {
      node_gemm_body<HP, EPI, PRE, 3, 1>(wb, Wa, sXa, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, pf, nextW, gPre);
    }
  }
}

// call-site form: the resident kernels call node_gemm directly (one more inlining level around it moves their instruction
// schedule: the compiled code of those kernels is kept identical to what was tuned and measured)
#define NODE_GEMM(EPI, ...)                                                   \
  do {                                                                        \
    if constexpr (GN != 0) node_gemm_n<HP, EPI, true, 3>(__VA_ARGS__);        \
    else node_gemm<HP, EPI, true>(__VA_ARGS__);                               \
  } while (0)

// Kernels with node buffers in global memory: a node GEMM's input rows are copied once into the idle weight ring of the
// edge GEMMs (LDS-DMA, 1 KiB per wave-instruction, no register round trip), so that the GEMM reads them from LDS like the
// resident kernels do -- per-lane global loads of the rows cost +39..55% per matrix (profiles/r04m).  `floats` is rounded
// up to whole 1 KiB units: the scratch and the ring leave that slack.
__device__ __forceinline__ void stage_rows(float* lds, const float* g, int floats, int wave, int lane) {
  const int units = (floats + 255) >> 8;
  for (int u = wave; u < units; u += kWaves)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + u * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(lds + u * 256), 16, 0, 0);
}
__device__ __forceinline__ void stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
__host__ __device__ __forceinline__ constexpr int stage_stride(int floats) { return (floats + 255) & ~255; }

// Per-molecule graph metadata prepared by the host (gaudi_hip.hip: build_meta8) and staged in LDS.
// Slots = the molecule's live edges sorted by receiving node, padded to whole 16-slot tiles; tile tau is served by wave
// tau & 7 in round tau >> 3.
struct MolGraph {
  int N, D, S;            // nodes (padded), 3+F, slot capacity of the batch (multiple of 16)
  int NC;                 // node columns that matter: 1 + last node that is live or touches a live edge (<= N)
  int ntiles, rounds;     // 16-slot tiles of THIS molecule, ceil(ntiles / 8)
  int pubx, pub_ch;       // predictor reverse pass: see w8_pred.h
  int hk = 0;             // float offset of the kept split copy of h from the start of LDS, 0 = none (w8_nodes_f16.h)
  const float* mask;      // LDS [N]
  const uint32_t* edge;   // LDS [S]  edge words (ew_* above)
  const float* em;        // LDS [S]  edge_mask value (0 for padding slots)
  const uint32_t* seg;    // LDS [N]  start << 16 | len : slots whose RECEIVING node is n
  const uint16_t* soff;   // LDS [N+1] CSR offsets into sidx: slots whose SENDING node is n ...
  const uint16_t* sidx;   // LDS [S]   ... ascending (= ascending receiving node)
  // packed launches: the graph is a disjoint union of up to 4 molecules (components); per-molecule reductions (centre of
  // gravity, gradient clip, NaN scrub) run per component.  row[n] = global row | component << 28, -1 = empty slot
  const int* row;         // LDS [N]
  int ncomp;
  int NR;                 // rows per molecule of the global arrays (N, or less when the group is wider than a molecule)
};
// pointer to float offset `off` of the workgroup's dynamic LDS (0 -> nullptr): the kept split copy of h sits behind the whole plan
__device__ __forceinline__ float* lds_at(int off) {
  extern __shared__ __attribute__((aligned(16))) float gaudi_lds_alias[];
  return off ? gaudi_lds_alias + off : nullptr;
}
__device__ __forceinline__ int mg_comp(const MolGraph& mg, int n) { return (mg.row[n] >> 28) & 7; }

}  // namespace w8
}  // namespace gaudi
