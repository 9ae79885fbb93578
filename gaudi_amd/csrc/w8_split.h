// w8_split.h -- edge-level GEMMs of the 8-wave kernels on the 16-bit matrix pipe with fp32-equivalent accuracy.
//
// v_mfma_f32_16x16x4_f32 runs at the vector fp32 rate (64 FLOP/clk/SIMD); the 16-bit instructions v_mfma_f32_16x16x32_{bf16,f16}
// at 16x that.  Round 2 split every fp32 operand into three bf16 pieces (six piece products per product: 2.67x less matrix
// time than the fp32 instruction).  Round 5: both operands are fp16 PAIRS, x s = hi + lo with hi = fp16(x s), lo = fp16(x s - hi)
// (round to nearest; s a power of two: per network for the weights, on the host -- the largest |w| of the network goes to
// [2^13, 2^14), gaudi_hip.hip: NodeScale --; per edge column for the activations, on the device -- the column's bound goes to [2^14, 2^15)), and a product is accumulated in fp32 from THREE piece products
//     a_hi b_lo + a_lo b_hi + a_hi b_hi                       (dropped: a_lo b_lo <= 2^-22 |a b|)
// on v_mfma_f32_16x16x32_f16: half the matrix instructions of the bf16 form, two thirds of its weight bytes (4 B per weight: the
// ring is 26 KiB per slot instead of 39) and a cheaper operand split (two conversions per value instead of three).  hi carries
// 11 significant bits, the remainder is exact in fp32 and lo carries its leading 11: 22 bits for every entry within 2^17 of its
// column's (its network's) largest, degrading to fp16's absolute floor 2^-24 (of a maximum of 2^14: 2^-38 relative to the
// largest entry) below that; piece products are exact in fp32.  Against float64 the result is as close as the fp32 instruction's
// (tests/test_gpu_split.py, tools/node_gemm_h_microbench.hip for the same arithmetic in the node GEMMs); unlike the node form
// (w8_nodes_f16.h: lo scaled by 2^11, two accumulators) the edge form keeps ONE accumulator per output tile -- a wave holds 13
// of them beside its operands.
// Range.  The column scale of a CHAINED GEMM is the exact maximum of the lane's own inputs (they are in registers); for the
// GEMM that generates its inputs chunk by chunk, silu(P_i + Q_j + c_r r + c_d d0), it comes from the bound
// |silu(u)| <= |u| <= max|P_i| + max|Q_j| + max|c_r| r + max|c_d| |d0| (row maxima from the node GEMMs' epilogues, column maxima
// from the host) -- a loose bound costs nothing, the pieces have 38 binades of room below the column's largest entry.
//
// Layout.  K is consumed in chunks of 32 inputs = two 16-feature tiles (2m, 2m+1); lane (column c, group g) carries inputs
// 16(2m) + 4g .. +3 in slots 0-3 and 16(2m+1) + 4g .. +3 in slots 4-7, which is exactly the accumulator layout of two output
// tiles of the previous GEMM of a chain.  The host packs each matrix as units of 1 KiB: unit (m, t, p) = piece p of output
// tile t against chunk m, lane L = (row L & 15, group L >> 4) holding its 8 fp16 A-operand slots (16 B): one conflict-free
// ds_read_b128 per unit.  Units are ordered [m][t][p]; a ring group = the CH tiles x 2 pieces one trip consumes.
#pragma once
#include "w8_common.h"
#ifndef GAUDI_EDGE_PRIO
#define GAUDI_EDGE_PRIO 0  // experiment (round 6): waves 4-7 at s_setprio 1 inside the trips -- bit 0: the generating edge GEMM, bit 1: the chained
                           // ones.  What pays in the node GEMMs' K loops LOSES here (same-session A/B, C3 mol/s): generating 208.3 -> 206.1 (C4
                           // 245.1 -> 244.5), chained 208.1 -> 206.2
#endif

// (The experiment variants of rounds 2-3 -- LDS-counter trips, register-staged ring, sliced input generation, L2 touches, cache
// policy bits -- live in tools/experiments/w8_split_variants.h, which only the microbenchmarks include; DESIGN.md section 8
// records what each measured.)

namespace gaudi {
namespace w8 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;
typedef __attribute__((ext_vector_type(2))) uint32_t u2;
constexpr int kPieces = 2;  // fp16 pieces per operand

template <int N, class F>
__device__ __forceinline__ void static_for(F f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

struct B3 {
  u4 h, l;  // the lane's 8 inputs of a K chunk, two fp16 pieces each (slot e in bits 16(e&1) of word e>>1)
};

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {  // v_cvt_pk_f16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a, b}, h2));
}
__device__ __forceinline__ f2 unpk_f16(uint32_t p) { return __builtin_convertvector(__builtin_bit_cast(h2, p), f2); }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// (by value: __builtin_bit_cast applied to an ext-vector element expression reads element 0 whatever the index -- hipcc 7.2)
__device__ __forceinline__ uint32_t absbits(float v) { return __builtin_bit_cast(uint32_t, v) & 0x7fffffffu; }
// power-of-two scale that brings a magnitude with these bits to [2^14, 2^15) (below fp16's 65 504), and its inverse; exponents
// clamped so that both factors are normal numbers
struct Pow2Scale {
  float s, inv;
};
__device__ __forceinline__ Pow2Scale scale_for(uint32_t maxbits) {
  int k = 141 - (int)(maxbits >> 23);
  k = k > 126 ? 126 : k;
  return Pow2Scale{__builtin_bit_cast(float, (uint32_t)(k + 127) << 23), __builtin_bit_cast(float, (uint32_t)(127 - k) << 23)};
}
struct P3 {
  uint32_t h, l;
};
__device__ __forceinline__ P3 split2(float a, float b) {  // (already scaled)
  P3 r;
  r.h = pk_f16(a, b);
  const f2 f = unpk_f16(r.h);
  r.l = pk_f16(a - f[0], b - f[1]);
  return r;
}
__device__ __forceinline__ B3 split8(const f4 lo, const f4 hi, float s) {
  // the scaled values on the VECTORS (packed multiplies, two values per instruction; exact: s is a power of two); the remainder as
  // fma(x, s, -hi) so that it stays ONE v_fma_mix_f32 that reads the fp16 piece directly (x * s - hi written on the packed product
  // costs a v_cvt_f32_f16 and a v_sub_f32 instead).  Same values either way.
  const f4 ls = lo * s, hs = hi * s;
  auto sp = [s](float a, float b, float as, float bs) {
    P3 r;
    r.h = pk_f16(as, bs);
    const f2 f = unpk_f16(r.h);
    r.l = pk_f16(__builtin_fmaf(a, s, -f[0]), __builtin_fmaf(b, s, -f[1]));
    return r;
  };
  const P3 p0 = sp(lo[0], lo[1], ls[0], ls[1]), p1 = sp(lo[2], lo[3], ls[2], ls[3]), p2 = sp(hi[0], hi[1], hs[0], hs[1]),
           p3 = sp(hi[2], hi[3], hs[2], hs[3]);
  B3 r;
  r.h = (u4){p0.h, p1.h, p2.h, p3.h};
  r.l = (u4){p0.l, p1.l, p2.l, p3.l};
  return r;
}
__device__ __forceinline__ f4 mfma_bf(const u4 a, const u4 b, const f4 c) {  // (the name is round 2's: the 16-bit matrix instruction)
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
// the largest |.| bits of the lane's column over its 4 lane groups (lanes c, c + 16, c + 32, c + 48)
__device__ __forceinline__ uint32_t column_max(uint32_t v) {
  v = umax(v, (uint32_t)__shfl_xor((int)v, 16));
  v = umax(v, (uint32_t)__shfl_xor((int)v, 32));
  return v;
}

// Geometry of a split matrix: T tiles, NC = ceil(T/2) K chunks, trips of CH output tiles (NH per chunk).
// MODE 1: a trip = a whole chunk (CH = T; ring = 2 x 3T KiB).  MODE 2: two trips per chunk (CH = ceil(T/2); ring = half of
// that, one more barrier per chunk): the form for molecules whose node buffers leave no room for the full ring.
template <int HP, int MODE>
struct SplitGeo {
  static_assert(MODE == 1 || MODE == 2, "split mode");
  static constexpr int T = HP / 16;
  static constexpr int NC = (T + 1) / 2;
  // a matrix needs at least two trips (the ring runs a trip ahead and knows only the current and the next matrix of a chain)
  static constexpr int CH = (MODE == 2 || NC == 1) && T > 1 ? (T + 1) / 2 : T;  // output tiles per trip
  static constexpr int NH = (T + CH - 1) / CH;
  static constexpr int kTrips = NC * NH;
  static constexpr int kUnit = 256;                    // floats (1 KiB)
  static constexpr int kSlotFloats = CH * kPieces * kUnit;   // LDS per ring slot
  static constexpr int kMatFloats = NC * T * kPieces * kUnit;  // one packed matrix
  static constexpr int UT = (CH * kPieces + kWaves - 1) / kWaves;
  // K tail (nf % 16 == 4, odd tile count: the last chunk is the tail tile alone): that chunk is stored as T fp32 tiles in
  // the K-tail form of w8_common.h (input 16(T-1)+g on lane group g, element 0) and issued as ONE trip of one fp32 k-step
  // per tile
  static constexpr bool kTailOK = (T & 1) && T >= 3;
  static constexpr int kTripsTail = (NC - 1) * NH + 1;
  static_assert(!kTailOK || T <= CH * kPieces, "the fp32 tail tiles must fit one ring slot");
  __host__ __device__ static constexpr int tiles_of(int h) { return (h + 1) * CH <= T ? CH : T - h * CH; }
};

template <int HP, int MODE>
struct RingS {
  using G = SplitGeo<HP, MODE>;
  float* base;  // LDS [2][kSlotFloats]
  int par;
  bool ktail;   // the matrices carry a K tail (SplitGeo::kTailOK widths only)
  const float* gbase;  // the split weight buffer (LDS-DMA loads take a plain global address)
  float winv;          // 2^-s of the network's weight images
  __device__ __forceinline__ float* slot(int p) const { return base + p * G::kSlotFloats; }
};

// Group of trip `tr` of a chain: trips 0 .. n-1 belong to the matrix at float offset W, trip n + k is trip k of nextW
// (nextW < 0: nothing follows).  n = kTrips, or kTripsTail when the matrices carry a K tail (their last chunk is ONE trip of
// T fp32 tiles).  -> float offset of the group, its 1 KiB units; false: nothing to load.
template <int HP, int MODE>
__device__ __forceinline__ bool trip_group(const RingS<HP, MODE>& r, int W, int nextW, int tr, int& off, int& units) {
  using G = SplitGeo<HP, MODE>;
  const bool tail = G::kTailOK && r.ktail;
  const int n = tail ? G::kTripsTail : G::kTrips;
  const bool nxt = tr >= n;
  const int t2 = nxt ? tr - n : tr;
  const int base = nxt ? nextW : W;
  if (tail && t2 == n - 1) {
    units = G::T;
    off = base + (G::NC - 1) * G::T * kPieces * G::kUnit;
  } else {
    const int m = t2 / G::NH, h = t2 % G::NH;
    units = G::tiles_of(h) * kPieces;
    off = base + (m * G::T + h * G::CH) * kPieces * G::kUnit;
  }
  off = __builtin_amdgcn_readfirstlane(off);
  units = __builtin_amdgcn_readfirstlane(units);
  // `units` is the same for every trip of a matrix: left visible, hipcc computes the "unit < units" tests of all the ring loads
  // once per GEMM chain, keeps them as lane masks across the chain and spills them (two v_readlane per ring load and trip)
  asm volatile("" : "+s"(units));
  return !(nxt && nextW < 0);
}
// LDS-DMA: unit un of the group goes straight to slot + un KiB (wave-uniform LDS base + 16 B per lane)
template <int HP, int MODE>
__device__ __forceinline__ bool rings_dma(const RingS<HP, MODE>& r, float* slot, int W, int nextW, int tr, int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  int off, units;
  if (!trip_group(r, W, nextW, tr, off, units)) return false;
#pragma unroll
  for (int u = 0; u < G::UT; ++u) {
    const int un = wave + kWaves * u;
    if (un < units)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(r.gbase + off + un * G::kUnit + lane * 4),
                                       (__attribute__((address_space(3))) void*)(slot + un * G::kUnit), 16, 0, 0);
  }
  return true;
}
template <int HP, int MODE>
__device__ __forceinline__ void rings_start(RingS<HP, MODE>& r, const WBuf&, int W, int wave, int lane) {
  rings_dma(r, r.slot(r.par), W, -1, 0, wave, lane);
}
// trip tr has passed its opening barrier: nobody reads slot(par ^ 1) any more; the group of trip tr + 1 must have landed
// at the next barrier (trip_barrier waits vmcnt(0))
template <int HP, int MODE>
__device__ __forceinline__ void rings_stage(RingS<HP, MODE>& r, const WBuf&, int W, int nextW, int tr, int wave, int lane) {
  rings_dma(r, r.slot(r.par ^ 1), W, nextW, tr + 1, wave, lane);
}

// The barrier that opens a trip.  With LDS-DMA the group this trip reads was written by global_load_lds instructions of ALL
// waves: each wave retires its own (vmcnt) before the barrier -- hipcc does not track these loads for the __syncthreads
// fence on every path (seen in the ISA: barriers with lgkmcnt(0) only, and a run-to-run difference in the results).
__device__ __forceinline__ void trip_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
// open a trip on slot par / close it (the slot is free for this wave)
template <int HP, int MODE>
__device__ __forceinline__ void trip_open(RingS<HP, MODE>& r, int lane) {
  (void)r;
  (void)lane;
  trip_barrier();
}
template <int HP, int MODE>
__device__ __forceinline__ void trip_close(RingS<HP, MODE>& r, int lane) {
  (void)lane;
  r.par ^= 1;
}

// One trip: NT output tiles (acc[t0 .. t0+NT)) against the K chunk in `b`; the A units of a tile are read one tile ahead of
// its MFMAs (explicit double buffer + fences: left alone, hipcc sinks every ds_read next to its MFMAs).  `mid` runs in the
// middle of the block (ring traffic, the next chunk's input generation).  The scheduling fence two tiles before the end of
// the trip is part of the tuned code shape (it is where the LDS-counter variant signalled; removing it moves instructions).
template <int HP, int MODE, int NT, bool ACT, class MID>
__device__ __forceinline__ void rings_mfma_act(f4* acc, const float* slot_lane, const B3& b, MID mid) {  // NOLINT
  constexpr int U = SplitGeo<HP, MODE>::kUnit;
  if constexpr (!ACT) {  // a wave without an edge tile in this round: its share of the ring traffic only
    mid();
    return;
  }
  // A units are read TWO tiles ahead (three register sets in rotation): a tile's three matrix instructions last 48 cycles, half of
  // what the six of the bf16 form did, and no longer cover a ds_read_b128 round trip on their own
  constexpr int kAhead = NT > 2 ? 2 : 1, kSets = kAhead + 1;
  f4 a[kSets][kPieces];
#pragma unroll
  for (int q = 0; q < kAhead; ++q)
#pragma unroll
    for (int p = 0; p < kPieces; ++p) a[q][p] = *(const f4*)(slot_lane + (q * kPieces + p) * U);
  static_for<NT>([&](auto t_tag) {
    constexpr int t = decltype(t_tag)::value;
    constexpr int cur = t % kSets;
    if (t == NT / 2) {
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t == (NT - 2 > NT / 2 ? NT - 2 : NT - 1)) __builtin_amdgcn_sched_barrier(0);
    if (t + kAhead < NT) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p) a[(t + kAhead) % kSets][p] = *(const f4*)(slot_lane + ((t + kAhead) * kPieces + p) * U);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const u4 ah = __builtin_bit_cast(u4, a[cur][0]), al = __builtin_bit_cast(u4, a[cur][1]);
      f4 c = acc[t];
      c = mfma_bf(al, b.h, c);  // small terms first
      c = mfma_bf(ah, b.l, c);
      c = mfma_bf(ah, b.h, c);
      acc[t] = c;
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}
// one wave-uniform branch per TRIP separates waves with and without a tile
template <int HP, int MODE, int NT, class MID>
__device__ __forceinline__ void rings_mfma(f4* acc, const float* slot_lane, const B3& b, bool active, MID mid) {  // NOLINT
  if (active) rings_mfma_act<HP, MODE, NT, true>(acc, slot_lane, b, mid);
  else rings_mfma_act<HP, MODE, NT, false>(acc, slot_lane, b, mid);
}

// The K-tail trip: one fp32 k-step per output tile (A = element 0 of the tile's lane-linear float4, B = the lane group's input)
template <int HP, int MODE, class MID>
__device__ __forceinline__ void rings_mfma_tail(f4 (&acc)[HP / 16], const float* slot_lane, float b, bool active, MID mid) {
  constexpr int T = HP / 16;
  float a[T];
#pragma unroll
  for (int t = 0; t < T; ++t) a[t] = slot_lane[t * SplitGeo<HP, MODE>::kUnit];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t == T / 2) {
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (active) acc[t] = mfma1(a[t], b, acc[t]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// acc = b2 + W2 . silu(u) (see edge_gemm_pq); weights in split format at float offset W of wb.  ubound >= |u| for every input of
// the lane's edge column (the caller's bound: row maxima of P_i and Q_j + column maxima of c_r, c_d times r, |d0|): the column's
// power-of-two scale.  The accumulators run in scaled units (column scale x network weight scale) and are descaled, bias added,
// at the end.
template <int HP, int MODE>
__device__ __forceinline__ void edge_gemm_pq_s(f4 (&acc)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb, int W, int nextW,
                                               const float* sB2, const float* sCr, const float* sCd, const float* pp,
                                               const float* qq, float r, float d0, float ubound, bool active, int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  constexpr int T = G::T;
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = splat(0.f);
  const Pow2Scale sc = scale_for(absbits(ubound));
  const float se = sc.s;
  auto gen = [&](int m) {  // split silu(u) of K chunk m (tiles 2m, 2m+1; an odd T leaves the upper half of the last chunk 0)
    const f4 lo = silu4v(edge_u(pp, qq, sCr, sCd, g, 2 * m, r, d0));
    const f4 hi = 2 * m + 1 < T ? silu4v(edge_u(pp, qq, sCr, sCd, g, 2 * m + 1 < T ? 2 * m + 1 : 0, r, d0)) : splat(0.f);
    return split8(lo, hi, se);
  };
  B3 bin = gen(0), nb = bin;
  static_assert(G::NH <= 3, "at most three trips per K chunk");
  const bool tail = G::kTailOK && ring.ktail;
#if GAUDI_EDGE_PRIO & 1
  if (wave >= kWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif
  auto chunk = [&](int m, const B3& bin, B3& nb) {  // consumes `bin`, generates the next chunk into `nb`
    const int mn = m + 1 < G::NC ? m + 1 : m;  // the chunk generated during this one (clamped at the end: no branch)
    auto trip = [&](auto h_tag) {
      constexpr int h = decltype(h_tag)::value;
      constexpr int NT = G::tiles_of(h);
      const int tr = m * G::NH + h;
      trip_open(ring, lane);
      rings_mfma<HP, MODE, NT>(acc + h * G::CH, ring.slot(ring.par) + lane * 4, bin, active, [&] {
        rings_stage(ring, wb, W, nextW, tr, wave, lane);
        // every wave generates the NEXT chunk in the middle of its block (vector work co-issues with 16-bit MFMAs; the
        // staggered placement of the fp32 form is 2 % slower here, generation right after the barrier 7 %)
        if (h == G::NH - 1) nb = gen(mn);
      });
      trip_close(ring, lane);
    };
    trip(std::integral_constant<int, 0>{});
    if constexpr (G::NH > 1) trip(std::integral_constant<int, 1>{});
    if constexpr (G::NH > 2) trip(std::integral_constant<int, 2>{});
  };
  const int full = tail ? G::NC - 1 : G::NC;  // one copy of the chunk body: the tail only shortens the rolled loop
  // two chunks per iteration with the roles of the two operand sets swapped: no copy of the generated pieces per trip
  int m = 0;
#pragma unroll 1
  for (; m + 1 < full; m += 2) {
    chunk(m, bin, nb);
    chunk(m + 1, nb, bin);
  }
  if (m < full) chunk(m, bin, nb);
  if constexpr (G::kTailOK) {
    if (tail) {  // (the tail tiles are fp32, scaled by the network's exponent on the host: the same units as the 16-bit part)
      trip_open(ring, lane);
      const float bt = silu_f(edge_u_tail(pp, qq, sCr, sCd, g, T, r, d0)[0]) * se;
      rings_mfma_tail<HP, MODE>(acc, ring.slot(ring.par) + lane * 4, bt, active,
                          [&] { rings_stage(ring, wb, W, nextW, G::kTripsTail - 1, wave, lane); });
      trip_close(ring, lane);
    }
  }
#if GAUDI_EDGE_PRIO & 1
  __builtin_amdgcn_s_setprio(0);
#endif
  const float dsc = sc.inv * ring.winv;
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = acc[t] * dsc + *(const f4*)(sB2 + 16 * t + 4 * g);
}

// Chained edge GEMM, input in registers (accumulator layout of the previous GEMM): out = bias + rowinit + W . in.  The column scale
// is the exact maximum of the lane's own inputs (all lane groups of the column).
template <int HP, int MODE>
__device__ __forceinline__ void edge_gemm_regs_s(f4 (&out)[HP / 16], const f4 (&in)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb,
                                                 int W, int nextW, const float* sBias, const float* rowinit, bool active,
                                                 int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  constexpr int T = G::T;
  const int g = lane >> 4;
  uint32_t mx = 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    out[t] = splat(0.f);
    mx = umax(mx, umax(umax(absbits(in[t][0]), absbits(in[t][1])), umax(absbits(in[t][2]), absbits(in[t][3]))));
  }
  const Pow2Scale sc = scale_for(column_max(mx));
  const float se = sc.s;
  static_assert(G::NH <= 3, "at most three trips per K chunk");
  const int c = lane & 15;
#if GAUDI_EDGE_PRIO & 2
  if (wave >= kWaves / 2) __builtin_amdgcn_s_setprio(1);
#endif
  auto chunk = [&](auto m_tag) {
    constexpr int m = decltype(m_tag)::value;
    const B3 bin = split8(in[2 * m], 2 * m + 1 < T ? in[2 * m + 1 < T ? 2 * m + 1 : 0] : splat(0.f), se);
    auto trip = [&](auto h_tag) {
      constexpr int h = decltype(h_tag)::value;
      constexpr int tr = m * G::NH + h;
      trip_open(ring, lane);
      rings_mfma<HP, MODE, G::tiles_of(h)>(out + h * G::CH, ring.slot(ring.par) + lane * 4, bin, active,
                                           [&] { rings_stage(ring, wb, W, nextW, tr, wave, lane); });
      trip_close(ring, lane);
    };
    trip(std::integral_constant<int, 0>{});
    if constexpr (G::NH > 1) trip(std::integral_constant<int, 1>{});
    if constexpr (G::NH > 2) trip(std::integral_constant<int, 2>{});
  };
  auto finish = [&] {  // descale, then bias and the per-row initial value
#if GAUDI_EDGE_PRIO & 2
    __builtin_amdgcn_s_setprio(0);
#endif
    const float dsc = sc.inv * ring.winv;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f4 b = sBias != nullptr ? *(const f4*)(sBias + 16 * t + 4 * g) : splat(0.f);
      if (rowinit != nullptr) b = b + *(const f4*)(rowinit + 16 * t + 4 * g);
      out[t] = out[t] * dsc + b;
    }
  };
  static_for<G::NC - 1>([&](auto m_tag) { chunk(m_tag); });
  if constexpr (G::kTailOK) {
    if (ring.ktail) {
      trip_open(ring, lane);
      rings_mfma_tail<HP, MODE>(out, ring.slot(ring.par) + lane * 4, tail_to_b(in[T - 1], c, g)[0] * se, active,
                          [&] { rings_stage(ring, wb, W, nextW, G::kTripsTail - 1, wave, lane); });
      trip_close(ring, lane);
      finish();
      return;
    }
  }
  chunk(std::integral_constant<int, G::NC - 1>{});
  finish();
}

// ---------------------------------------------------------------------------------------------
// One interface over both edge-GEMM engines.  Matrix offsets are the fp32 weight buffer's float offsets
// in both cases: the split image of the matrix at float offset W lives at float offset 2 W of its own buffer (an image is
// the size of the fp32 matrix, for an odd tile count up to 1.08x; the holes are never touched).
// ---------------------------------------------------------------------------------------------
template <int HP, int SP>  // SP: 0 = fp32 matrix instructions, 1 / 2 = split operands with the full / half ring (SplitGeo)
struct EdgeRing {
  using type = RingS<HP, SP>;
  static constexpr int kFloats = 2 * SplitGeo<HP, SP>::kSlotFloats;
};
template <int HP>
struct EdgeRing<HP, 0> {
  using type = Ring<HP>;
  static constexpr int kFloats = 2 * (HP / 16) * 256;
};
__host__ __device__ constexpr int edge_ring_floats(int HP, int mode) {
  const int T = HP / 16;
  const int CH = (mode == 2 || (T + 1) / 2 == 1) && T > 1 ? (T + 1) / 2 : T;  // = SplitGeo<HP, mode>::CH
  return mode ? 2 * CH * kPieces * 256 : 2 * T * 256;
}
static_assert(edge_ring_floats(32, 1) == EdgeRing<32, 1>::kFloats && edge_ring_floats(208, 1) == EdgeRing<208, 1>::kFloats &&
                  edge_ring_floats(48, 2) == EdgeRing<48, 2>::kFloats && edge_ring_floats(208, 2) == EdgeRing<208, 2>::kFloats,
              "host LDS planning and SplitGeo disagree");
__device__ __forceinline__ int split_off(int W) { return W < 0 ? -1 : 2 * W; }

template <int HP>
__device__ __forceinline__ void er_init(Ring<HP>& r, float* base, bool ktail, const float*, float) {
  r.base = base;
  r.par = 0;
  r.ktail = ktail;
}
template <int HP, int MODE>
__device__ __forceinline__ void er_init(RingS<HP, MODE>& r, float* base, bool ktail, const float* ws, float winv) {
  r.base = base;
  r.par = 0;
  r.ktail = ktail;
  r.gbase = ws;
  r.winv = winv;
}
template <int HP>
__device__ __forceinline__ void er_start(Ring<HP>& r, const WBuf& wb, int W, int wave, int lane) {
  ring_start<HP>(r, wb, W, wave, lane);
}
template <int HP, int MODE>
__device__ __forceinline__ void er_start(RingS<HP, MODE>& r, const WBuf& wb, int W, int wave, int lane) {
  rings_start(r, wb, split_off(W), wave, lane);
}
// ubound: an upper bound of |u| over the inputs of the lane's edge (split form: the column's scale; unused by the fp32 form)
template <int HP>
__device__ __forceinline__ void er_gemm_pq(f4 (&acc)[HP / 16], Ring<HP>& ring, const WBuf& wb, int W, int nextW, const float* sB2,
                                           const float* sCr, const float* sCd, const float* pp, const float* qq, float r, float d0,
                                           float, bool active, int wave, int lane STAMP_DECL) {
  edge_gemm_pq<HP>(acc, ring, wb, W, nextW, sB2, sCr, sCd, pp, qq, r, d0, active, wave, lane STAMP_ARGS);
}
template <int HP, int MODE>
__device__ __forceinline__ void er_gemm_pq(f4 (&acc)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb, int W, int nextW, const float* sB2,
                                           const float* sCr, const float* sCd, const float* pp, const float* qq, float r, float d0,
                                           float ubound, bool active, int wave, int lane STAMP_DECL) {
  edge_gemm_pq_s(acc, ring, wb, split_off(W), split_off(nextW), sB2, sCr, sCd, pp, qq, r, d0, ubound, active, wave, lane);
}
template <int HP>
__device__ __forceinline__ void er_gemm_regs(f4 (&out)[HP / 16], const f4 (&in)[HP / 16], Ring<HP>& ring, const WBuf& wb, int W,
                                             int nextW, const float* sBias, const float* rowinit, bool active, int wave, int lane) {
  edge_gemm_regs<HP>(out, in, ring, wb, W, nextW, sBias, rowinit, active, wave, lane);
}
template <int HP, int MODE>
__device__ __forceinline__ void er_gemm_regs(f4 (&out)[HP / 16], const f4 (&in)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb, int W,
                                             int nextW, const float* sBias, const float* rowinit, bool active, int wave, int lane) {
  edge_gemm_regs_s(out, in, ring, wb, split_off(W), split_off(nextW), sBias, rowinit, active, wave, lane);
}

}  // namespace w8
}  // namespace gaudi
