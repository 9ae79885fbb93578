// w8_split_variants.h -- the split-operand edge GEMM of gaudi_amd/csrc/w8_split.h WITH the experiment knobs of rounds 2-3
// (DESIGN.md section 8): GAUDI_TRIP_FLAGS (LDS counters instead of the trip barrier), GAUDI_RING_PREFETCH (L2 touch two trips
// ahead), GAUDI_SPLIT_GLDS=0 (register-staged ring), MB_NO_BARRIER, GAUDI_SPLIT_SGB, GAUDI_RING_AUX, GAUDI_SPLIT_GENSPREAD,
// GAUDI_TRIP_LATE, GAUDI_SPLIT_PD, GAUDI_SPLIT_PQ_UNROLL2=0, GAUDI_RING_OPAQUE_UNITS=0.  Every one of them was measured and
// rejected (or folded into the default); the production header carries the default form only.  Included by
// tools/split_gemm_microbench.hip INSTEAD of w8_split.h (same namespace, same names) -- never by the library.
//
// v_mfma_f32_16x16x4_f32 runs at the vector fp32 rate (64 FLOP/clk/SIMD); v_mfma_f32_16x16x32_bf16 at 16x that.  Every fp32
// operand is split into three bf16 pieces (round-to-nearest each time: x = h + m + l EXACTLY, |m| <= 2^-9 |x|, |l| <= 2^-17
// |x|) and a product a.b is accumulated in fp32 from the six piece products of weight >= 2^-18:
//     ah.bh + ah.bm + am.bh + am.bm + ah.bl + al.bh          (dropped: am.bl + al.bm + al.bl <= 2^-24.4 |a.b|)
// Piece products are exact in fp32 (8 x 8 significant bits), so the only error beside the fp32 accumulation is the dropped
// tail -- below the rounding error of ONE fp32 multiply (2^-24 |a.b|).  Measured against float64 the result is about as close as
// the fp32-MFMA GEMM (tests/test_gpu_split.py, tools/split_gemm_microbench.hip).  Six 16-cycle instructions replace eight
// 32-cycle ones per 32 inputs: 2.67x less matrix time, and vector-ALU work co-issues with bf16 MFMAs (it does not with
// fp32 MFMAs).
//
// Layout.  K is consumed in chunks of 32 inputs = two 16-feature tiles (2m, 2m+1); lane (column c, group g) carries inputs
// 16(2m) + 4g .. +3 in slots 0-3 and 16(2m+1) + 4g .. +3 in slots 4-7, which is exactly the accumulator layout of two output
// tiles of the previous GEMM of a chain.  The host packs each matrix as units of 1 KiB: unit (m, t, p) = piece p of output
// tile t against chunk m, lane L = (row L & 15, group L >> 4) holding its 8 bf16 A-operand slots (16 B): one conflict-free
// ds_read_b128 per unit.  Units are ordered [m][t][p]; a ring group = the CH tiles x 3 pieces one trip consumes.
#pragma once
#include "w8_common.h"

#ifndef GAUDI_SPLIT_GLDS
#define GAUDI_SPLIT_GLDS 1  // 1: the ring is filled by global_load_lds_dwordx4 (no staging registers, no ds_write); 0: register staging
#endif

#ifndef GAUDI_RING_AUX
#define GAUDI_RING_AUX 0  // cache-policy bits of the LDS-DMA ring loads (experiment knob: 2 = nt)
#endif

#ifndef GAUDI_RING_OPAQUE_UNITS
#define GAUDI_RING_OPAQUE_UNITS 1
#endif
#ifndef GAUDI_TRIP_FLAGS
#define GAUDI_TRIP_FLAGS 0  // 1: trips are opened by per-slot FULL / FREE counters in LDS instead of a workgroup barrier (experiment:
                            // correct and slower; exercised by tools/split_gemm_microbench.hip only)
#endif

namespace gaudi {
namespace w8 {

#if GAUDI_TRIP_FLAGS
// Split barrier of the weight ring.  Four counters in LDS: F[s] = waves whose share of a fill of slot s has landed,
// D[s] = waves that finished reading a fill of slot s.  They only grow; every wave keeps use[s] = completed uses of slot s
// (the control flow is workgroup-uniform), so the k-th fill of slot s is complete at F[s] = 8 k and free again at D[s] = 8 k.
// A wave SIGNALS as early as it can (its LDS-DMA retired: before the last tile of the trip) and WAITS as late as it must
// (FULL when it opens the next trip, FREE in the middle of a trip before it overwrites the other slot): waves may drift
// apart by that much instead of meeting at a barrier every trip.  LDS operations of a wave execute in order, so a
// counter update follows the wave's earlier reads and a read after a successful poll sees the other waves' data.
typedef __attribute__((address_space(3))) int lds_int;  // LDS address space: ds_read / ds_add, not flat instructions (a flat
                                                        // load would also wait for the wave's LDS-DMA loads in flight)
__device__ __forceinline__ void flag_wait(const lds_int* f, int target) {
  int v;
  do {
    v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  } while (v < target);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void flag_add(lds_int* f, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

template <int N, class F>
__device__ __forceinline__ void static_for(F f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

struct B3 {
  u4 h, m, l;  // the lane's 8 inputs of a K chunk, three bf16 pieces each (slot e in bits 16(e&1) of word e>>1)
};

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a, b}, bf2));
}
struct P3 {
  uint32_t h, m, l;
};
__device__ __forceinline__ P3 split2(float a, float b) {
  P3 r;
  r.h = pk_bf16(a, b);
  const float ra = a - __builtin_bit_cast(float, r.h << 16), rb = b - __builtin_bit_cast(float, r.h & 0xffff0000u);
  r.m = pk_bf16(ra, rb);
  r.l = pk_bf16(ra - __builtin_bit_cast(float, r.m << 16), rb - __builtin_bit_cast(float, r.m & 0xffff0000u));
  return r;
}
__device__ __forceinline__ B3 split8(const f4 lo, const f4 hi) {
  const P3 p0 = split2(lo[0], lo[1]), p1 = split2(lo[2], lo[3]), p2 = split2(hi[0], hi[1]), p3 = split2(hi[2], hi[3]);
  B3 r;
  r.h = (u4){p0.h, p1.h, p2.h, p3.h};
  r.m = (u4){p0.m, p1.m, p2.m, p3.m};
  r.l = (u4){p0.l, p1.l, p2.l, p3.l};
  return r;
}
__device__ __forceinline__ f4 mfma_bf(const u4 a, const u4 b, const f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
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
  static constexpr int kSlotFloats = CH * 3 * kUnit;   // LDS per ring slot
  static constexpr int kMatFloats = NC * T * 3 * kUnit;  // one packed matrix
  static constexpr int UT = (CH * 3 + kWaves - 1) / kWaves;
  // K tail (nf % 16 == 4, odd tile count: the last chunk is the tail tile alone): that chunk is stored as T fp32 tiles in
  // the K-tail form of w8_common.h (input 16(T-1)+g on lane group g, element 0) and issued as ONE trip of one fp32 k-step
  // per tile
  static constexpr bool kTailOK = (T & 1) && T >= 3;
  static constexpr int kTripsTail = (NC - 1) * NH + 1;
  static_assert(!kTailOK || T <= CH * 3, "the fp32 tail tiles must fit one ring slot");
  __host__ __device__ static constexpr int tiles_of(int h) { return (h + 1) * CH <= T ? CH : T - h * CH; }
};

template <int HP, int MODE>
struct RingS {
  using G = SplitGeo<HP, MODE>;
  float* base;  // LDS [2][kSlotFloats]
  int par;
  bool ktail;   // the matrices carry a K tail (SplitGeo::kTailOK widths only)
#if GAUDI_SPLIT_GLDS
  const float* gbase;  // the split weight buffer (LDS-DMA loads take a plain global address)
#else
  f4 st[G::UT];
#endif
#if GAUDI_TRIP_FLAGS
  lds_int* fl;   // LDS: F[0], F[1], D[0], D[1]
  int use0, use1;  // completed uses of each slot (two scalars: a dynamically indexed array would live in scratch)
  bool pend;     // this wave issued a fill of slot par ^ 1 (or, before the first trip, of slot par) and has not signalled it
  int pend_slot;
#endif
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
    off = base + (G::NC - 1) * G::T * 3 * G::kUnit;
  } else {
    const int m = t2 / G::NH, h = t2 % G::NH;
    units = G::tiles_of(h) * 3;
    off = base + (m * G::T + h * G::CH) * 3 * G::kUnit;
  }
  off = __builtin_amdgcn_readfirstlane(off);
  units = __builtin_amdgcn_readfirstlane(units);
#if GAUDI_RING_OPAQUE_UNITS
  // `units` is the same for every trip of a matrix: left visible, hipcc computes the "unit < units" tests of all the ring loads
  // once per GEMM chain, keeps them as lane masks across the chain and spills them (two v_readlane per ring load and trip)
  asm volatile("" : "+s"(units));
#endif
  return !(nxt && nextW < 0);
}
#if GAUDI_SPLIT_GLDS
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
                                       (__attribute__((address_space(3))) void*)(slot + un * G::kUnit), 16, 0, GAUDI_RING_AUX);
  }
  return true;
}
#if GAUDI_TRIP_FLAGS
// fill slot s for its next use: wait until every wave has finished reading its previous contents, then issue this wave's share
template <int HP, int MODE>
__device__ __forceinline__ void rings_fill(RingS<HP, MODE>& r, int s, int W, int nextW, int tr, int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  int off, units;
  if (!trip_group(r, W, nextW, tr, off, units)) return;
  flag_wait(r.fl + 2 + s, kWaves * (s ? r.use1 : r.use0));
  rings_dma(r, r.slot(s), W, nextW, tr, wave, lane);
  r.pend = true;
  r.pend_slot = s;
}
// this wave's share of the pending fill has landed: tell the others
template <int HP, int MODE>
__device__ __forceinline__ void rings_signal(RingS<HP, MODE>& r, int lane) {
  if (r.pend) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_add(r.fl + r.pend_slot, lane);
    r.pend = false;
  }
}
#endif
template <int HP, int MODE>
__device__ __forceinline__ void rings_start(RingS<HP, MODE>& r, const WBuf&, int W, int wave, int lane) {
#if GAUDI_TRIP_FLAGS
  rings_fill(r, r.par, W, -1, 0, wave, lane);
#else
  rings_dma(r, r.slot(r.par), W, -1, 0, wave, lane);
#endif
}
#ifndef GAUDI_RING_PREFETCH
#define GAUDI_RING_PREFETCH 0  // experiment: touch the group of trip tr + 2 (one dword per 128-byte line) so that its LDS-DMA hits L2
#endif
// trip tr has passed its opening barrier: nobody reads slot(par ^ 1) any more; the group of trip tr + 1 must have landed
// at the next barrier (trip_barrier waits vmcnt(0))
template <int HP, int MODE>
__device__ __forceinline__ void rings_stage(RingS<HP, MODE>& r, const WBuf&, int W, int nextW, int tr, int wave, int lane) {
#if GAUDI_TRIP_FLAGS
  rings_fill(r, r.par ^ 1, W, nextW, tr + 1, wave, lane);
#else
  rings_dma(r, r.slot(r.par ^ 1), W, nextW, tr + 1, wave, lane);
#endif
#if GAUDI_RING_PREFETCH
  // The weight set (58 MB) streams through a 4 MiB L2 once per step: the first CU of an XCD to ask for a line waits for the
  // Infinity Cache.  Each wave touches 64 lines (8 KiB) of the group after next; the value is discarded.
  {
    int off, units;
    if (trip_group(r, W, nextW, tr + 2, off, units)) {
      const int line = wave * 64 + lane;  // 128-byte lines of the group
      if (line < units * 8) {
        float dummy;
        const float* p = r.gbase + off + line * 32;
        asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(p) : "memory");
      }
    }
  }
#endif
}
#else
// register staging (kept for comparison: tools/split_gemm_microbench.hip): loads issued two trips ahead, stored mid-trip
template <int HP, int MODE>
__device__ __forceinline__ void rings_issue(RingS<HP, MODE>& r, const WBuf& wb, int W, int nextW, int tr, int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  int off, units;
  const bool have = trip_group(r, W, nextW, tr, off, units);
#pragma unroll
  for (int u = 0; u < G::UT; ++u) {
    const int un = wave + kWaves * u;
    r.st[u] = ldw4(wb, (have ? off : 0) + (un < units ? un : 0) * G::kUnit, (have && un < units) ? lane : kOOBLane);
  }
}
template <int HP, int MODE>
__device__ __forceinline__ void rings_commit(const RingS<HP, MODE>& r, float* slot, int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
#pragma unroll
  for (int u = 0; u < G::UT; ++u) {
    const int un = wave + kWaves * u;
    if (un < G::CH * 3) *(f4*)(slot + un * G::kUnit + lane * 4) = r.st[u];
  }
}
template <int HP, int MODE>
__device__ __forceinline__ void rings_start(RingS<HP, MODE>& r, const WBuf& wb, int W, int wave, int lane) {
  rings_issue(r, wb, W, -1, 0, wave, lane);
  rings_commit(r, r.slot(r.par), wave, lane);
  rings_issue(r, wb, W, -1, 1, wave, lane);
}
template <int HP, int MODE>
__device__ __forceinline__ void rings_stage(RingS<HP, MODE>& r, const WBuf& wb, int W, int nextW, int tr, int wave, int lane) {
  rings_commit(r, r.slot(r.par ^ 1), wave, lane);
  rings_issue(r, wb, W, nextW, tr + 2, wave, lane);
}
#endif

// The barrier that opens a trip.  With LDS-DMA the group this trip reads was written by global_load_lds instructions of ALL
// waves: each wave retires its own (vmcnt) before the barrier -- hipcc does not track these loads for the __syncthreads
// fence on every path (seen in the ISA: barriers with lgkmcnt(0) only, and a run-to-run difference in the results).
__device__ __forceinline__ void trip_barrier() {
#ifdef MB_NO_BARRIER  // timing experiment only (tools/split_gemm_microbench.hip): results are wrong without it
  return;
#endif
#if GAUDI_SPLIT_GLDS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
}
// open a trip on slot par / close it (the slot is free for this wave)
template <int HP, int MODE>
__device__ __forceinline__ void trip_open(RingS<HP, MODE>& r, int lane) {
#if GAUDI_TRIP_FLAGS
  rings_signal(r, lane);
  flag_wait(r.fl + r.par, kWaves * ((r.par ? r.use1 : r.use0) + 1));
#else
  (void)r;
  (void)lane;
  trip_barrier();
#endif
}
template <int HP, int MODE>
__device__ __forceinline__ void trip_close(RingS<HP, MODE>& r, int lane) {
#if GAUDI_TRIP_FLAGS
  flag_add(r.fl + 2 + r.par, lane);
  r.use0 += r.par ^ 1;
  r.use1 += r.par;
#else
  (void)lane;
#endif
  r.par ^= 1;
}

// One trip: NT output tiles (acc[t0 .. t0+NT)) against the K chunk in `b`; A units one tile ahead in registers.
#ifndef GAUDI_SPLIT_PD
#define GAUDI_SPLIT_PD 1  // A units are read this many output tiles ahead of their MFMAs
#endif
#ifndef GAUDI_SPLIT_SGB
#define GAUDI_SPLIT_SGB 0  // > 0: sched_group_barrier pattern MFMA | this many vector instructions, per tile (experiment)
#endif
struct NoHook {
  template <class TT>
  __device__ __forceinline__ void operator()(TT) const {}
};
// `hook(tile tag)` runs next to the MFMAs of every output tile (inside the same scheduling region): vector work cut into
// small slices co-issues with the bf16 matrix instructions of BOTH waves of the SIMD instead of stalling the wave's own
// MFMA stream in one lump (GAUDI_SPLIT_GENSPREAD, edge_gemm_pq_s)
#ifndef GAUDI_SPLIT_PQ_UNROLL2
#define GAUDI_SPLIT_PQ_UNROLL2 1
#endif
#ifndef GAUDI_TRIP_LATE
#define GAUDI_TRIP_LATE 2  // the FULL signal of the fill issued mid-trip is given this many tiles before the end of the trip
#endif
template <int HP, int MODE, int NT, bool ACT, class MID, class HOOK, class LATE>
__device__ __forceinline__ void rings_mfma_act(f4* acc, const float* slot_lane, const B3& b, MID mid, HOOK hook, LATE late) {  // NOLINT
  constexpr int U = SplitGeo<HP, MODE>::kUnit;
  constexpr int PD = GAUDI_SPLIT_PD, NB = PD + 1;
  if constexpr (!ACT) {  // a wave without an edge tile in this round: its share of the ring traffic only
    mid();
    late();
    return;
  }
  f4 a[NB][3];
#pragma unroll
  for (int d = 0; d < PD; ++d)
    if (d < NT) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[d][p] = *(const f4*)(slot_lane + (d * 3 + p) * U);
    }
  static_for<NT>([&](auto t_tag) {
    constexpr int t = decltype(t_tag)::value;
    constexpr int cur = t % NB;
    if (t == NT / 2) {
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t == (NT - GAUDI_TRIP_LATE > NT / 2 ? NT - GAUDI_TRIP_LATE : NT - 1)) {
      late();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + PD < NT) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[(t + PD) % NB][p] = *(const f4*)(slot_lane + ((t + PD) * 3 + p) * U);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const u4 ah = __builtin_bit_cast(u4, a[cur][0]), am = __builtin_bit_cast(u4, a[cur][1]), al = __builtin_bit_cast(u4, a[cur][2]);
      f4 c = acc[t];
      c = mfma_bf(al, b.h, c);  // small terms first
      c = mfma_bf(ah, b.l, c);
      c = mfma_bf(am, b.m, c);
      c = mfma_bf(am, b.h, c);
      c = mfma_bf(ah, b.m, c);
      c = mfma_bf(ah, b.h, c);
      acc[t] = c;
    }
    hook(t_tag);
#if GAUDI_SPLIT_SGB
    // ask the scheduler for MFMA | 2 vector instructions | MFMA | ... inside this tile's region
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, GAUDI_SPLIT_SGB, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
  });
}
// `hook(tile tag)` runs next to the MFMAs of every output tile (inside the same scheduling region): vector work cut into
// small slices co-issues with the bf16 matrix instructions instead of stalling the wave's MFMA stream in one lump
// (GAUDI_SPLIT_GENSPREAD, edge_gemm_pq_s).  One wave-uniform branch per TRIP separates waves with and without a tile, so a
// tile's MFMAs and its slice share a basic block.
struct NoLate {
  __device__ __forceinline__ void operator()() const {}
};
template <int HP, int MODE, int NT, class MID, class HOOK = NoHook, class LATE = NoLate>
__device__ __forceinline__ void rings_mfma(f4* acc, const float* slot_lane, const B3& b, bool active, MID mid, HOOK hook = HOOK{},
                                           LATE late = LATE{}) {  // NOLINT
  if (active) rings_mfma_act<HP, MODE, NT, true>(acc, slot_lane, b, mid, hook, late);
  else rings_mfma_act<HP, MODE, NT, false>(acc, slot_lane, b, mid, hook, late);
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

#ifndef GAUDI_SPLIT_GENSPREAD
#define GAUDI_SPLIT_GENSPREAD 0  // 1: the next chunk's input generation runs in slices beside the tiles' MFMAs (measured: no gain -- the
                                 // vector issue port is as busy as the matrix pipe either way); 0: one block mid-trip
#endif
// The input generation of one K chunk (silu(u) of 8 inputs per lane, split into 3 bf16 pieces each) as kGenStages slices
// of ~10 vector instructions
struct GenPipe {
  static constexpr int kStages = 11;
  f4 p, q, cr, cd;   // operands of the sub-tile being assembled
  f4 ulo, uhi;       // u, then silu(u)
  B3 out;            // the three bf16 pieces of the 8 inputs
};
template <int S>
__device__ __forceinline__ void gen_stage(GenPipe& gp, const float* pp, const float* qq, const float* sCr, const float* sCd, int g,
                                          int T, int m, float r, float d0) {
  const int tl = 2 * m, th = 2 * m + 1 < T ? 2 * m + 1 : 0;  // an odd T leaves the upper half of the last chunk 0
  const bool has_hi = 2 * m + 1 < T;
  auto load = [&](int cc) {
    gp.p = *(const f4*)(pp + 16 * cc);
    gp.q = *(const f4*)(qq + 16 * cc);
    gp.cr = *(const f4*)(sCr + 16 * cc + 4 * g);
    gp.cd = *(const f4*)(sCd + 16 * cc + 4 * g);
  };
  auto u_of = [&] { return gp.p + gp.q + gp.cr * r + gp.cd * d0; };  // = edge_u (device_common.h), same operation order
  if constexpr (S == 0) load(tl);
  if constexpr (S == 1) { gp.ulo = u_of(); load(th); }
  if constexpr (S == 2) gp.uhi = has_hi ? u_of() : splat(0.f);
  if constexpr (S == 3) { gp.ulo[0] = silu_f(gp.ulo[0]); gp.ulo[1] = silu_f(gp.ulo[1]); }
  if constexpr (S == 4) { gp.ulo[2] = silu_f(gp.ulo[2]); gp.ulo[3] = silu_f(gp.ulo[3]); }
  if constexpr (S == 5) { gp.uhi[0] = silu_f(gp.uhi[0]); gp.uhi[1] = silu_f(gp.uhi[1]); }
  if constexpr (S == 6) { gp.uhi[2] = silu_f(gp.uhi[2]); gp.uhi[3] = silu_f(gp.uhi[3]); }
  if constexpr (S >= 7 && S <= 10) {
    constexpr int k = S - 7;  // word k of the B operand: inputs 2k, 2k+1
    const P3 t = k < 2 ? split2(gp.ulo[2 * (k & 1)], gp.ulo[2 * (k & 1) + 1]) : split2(gp.uhi[2 * (k & 1)], gp.uhi[2 * (k & 1) + 1]);
    gp.out.h[k] = t.h;
    gp.out.m[k] = t.m;
    gp.out.l[k] = t.l;
  }
}
__device__ __forceinline__ B3 gen_result(const GenPipe& gp) { return gp.out; }

// acc = b2 + W2 . silu(u) (see edge_gemm_pq); weights in split format at float offset W of wb
template <int HP, int MODE>
__device__ __forceinline__ void edge_gemm_pq_s(f4 (&acc)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb, int W, int nextW,
                                               const float* sB2, const float* sCr, const float* sCd, const float* pp,
                                               const float* qq, float r, float d0, bool active, int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  constexpr int T = G::T;
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = *(const f4*)(sB2 + 16 * t + 4 * g);
  auto gen = [&](int m) {  // split silu(u) of K chunk m (tiles 2m, 2m+1; an odd T leaves the upper half of the last chunk 0)
    const f4 lo = silu4(edge_u(pp, qq, sCr, sCd, g, 2 * m, r, d0));
    const f4 hi = 2 * m + 1 < T ? silu4(edge_u(pp, qq, sCr, sCd, g, 2 * m + 1 < T ? 2 * m + 1 : 0, r, d0)) : splat(0.f);
    return split8(lo, hi);
  };
  B3 bin = gen(0), nb = bin;
  static_assert(G::NH <= 3, "at most three trips per K chunk");
  const bool tail = G::kTailOK && ring.ktail;
  auto chunk = [&](int m, const B3& bin, B3& nb) {  // consumes `bin`, generates the next chunk into `nb`
    const int mn = m + 1 < G::NC ? m + 1 : m;  // the chunk generated during this one (clamped at the end: no branch)
    auto trip = [&](auto h_tag) {
      constexpr int h = decltype(h_tag)::value;
      constexpr int NT = G::tiles_of(h);
      const int tr = m * G::NH + h;
      trip_open(ring, lane);
#if GAUDI_SPLIT_GENSPREAD
      if constexpr (h == G::NH - 1) {
        // stage s of the generation runs beside tile s * NT / kStages ... spread evenly over the trip's tiles
        GenPipe gp;
        rings_mfma<HP, MODE, NT>(
            acc + h * G::CH, ring.slot(ring.par) + lane * 4, bin, active, [&] { rings_stage(ring, wb, W, nextW, tr, wave, lane); },
            [&](auto t_tag) {
              constexpr int t = decltype(t_tag)::value;
              constexpr int s0 = t * GenPipe::kStages / NT, s1 = (t + 1) * GenPipe::kStages / NT;
              static_for<s1 - s0>([&](auto k_tag) { gen_stage<s0 + decltype(k_tag)::value>(gp, pp, qq, sCr, sCd, g, T, mn, r, d0); });
            });
        nb = gen_result(gp);
      } else {
        rings_mfma<HP, MODE, NT>(acc + h * G::CH, ring.slot(ring.par) + lane * 4, bin, active,
                                 [&] { rings_stage(ring, wb, W, nextW, tr, wave, lane); });
      }
#else
      rings_mfma<HP, MODE, NT>(acc + h * G::CH, ring.slot(ring.par) + lane * 4, bin, active, [&] {
        rings_stage(ring, wb, W, nextW, tr, wave, lane);
        // every wave generates the NEXT chunk in the middle of its block (vector work co-issues with bf16 MFMAs; the
        // staggered placement of the fp32 form is 2 % slower here, generation right after the barrier 7 %)
        if (h == G::NH - 1) nb = gen(mn);
      }, NoHook{}, [&] {
#if GAUDI_TRIP_FLAGS
        rings_signal(ring, lane);
#endif
      });
#endif
      trip_close(ring, lane);
    };
    trip(std::integral_constant<int, 0>{});
    if constexpr (G::NH > 1) trip(std::integral_constant<int, 1>{});
    if constexpr (G::NH > 2) trip(std::integral_constant<int, 2>{});
  };
  const int full = tail ? G::NC - 1 : G::NC;  // one copy of the chunk body: the tail only shortens the rolled loop
#if GAUDI_SPLIT_PQ_UNROLL2
  // two chunks per iteration with the roles of the two operand sets swapped: no copy of the generated pieces per trip
  int m = 0;
#pragma unroll 1
  for (; m + 1 < full; m += 2) {
    chunk(m, bin, nb);
    chunk(m + 1, nb, bin);
  }
  if (m < full) chunk(m, bin, nb);
#else
#pragma unroll 1
  for (int m = 0; m < full; ++m) {
    chunk(m, bin, nb);
    bin = nb;
  }
#endif
  if constexpr (G::kTailOK) {
    if (tail) {
      trip_open(ring, lane);
      const float bt = silu_f(edge_u_tail(pp, qq, sCr, sCd, g, T, r, d0)[0]);
      rings_mfma_tail<HP, MODE>(acc, ring.slot(ring.par) + lane * 4, bt, active,
                          [&] { rings_stage(ring, wb, W, nextW, G::kTripsTail - 1, wave, lane); });
      trip_close(ring, lane);
    }
  }
}

// Chained edge GEMM, input in registers (accumulator layout of the previous GEMM): out = bias + rowinit + W . in
template <int HP, int MODE>
__device__ __forceinline__ void edge_gemm_regs_s(f4 (&out)[HP / 16], const f4 (&in)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb,
                                                 int W, int nextW, const float* sBias, const float* rowinit, bool active,
                                                 int wave, int lane) {
  using G = SplitGeo<HP, MODE>;
  constexpr int T = G::T;
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f4 b = sBias != nullptr ? *(const f4*)(sBias + 16 * t + 4 * g) : splat(0.f);
    if (rowinit != nullptr) b = b + *(const f4*)(rowinit + 16 * t + 4 * g);
    out[t] = b;
  }
  static_assert(G::NH <= 3, "at most three trips per K chunk");
  const int c = lane & 15;
  auto chunk = [&](auto m_tag) {
    constexpr int m = decltype(m_tag)::value;
    const B3 bin = split8(in[2 * m], 2 * m + 1 < T ? in[2 * m + 1 < T ? 2 * m + 1 : 0] : splat(0.f));
    auto trip = [&](auto h_tag) {
      constexpr int h = decltype(h_tag)::value;
      constexpr int tr = m * G::NH + h;
      trip_open(ring, lane);
      rings_mfma<HP, MODE, G::tiles_of(h)>(out + h * G::CH, ring.slot(ring.par) + lane * 4, bin, active,
                                           [&] { rings_stage(ring, wb, W, nextW, tr, wave, lane); }, NoHook{}, [&] {
#if GAUDI_TRIP_FLAGS
                                             rings_signal(ring, lane);
#endif
                                           });
      trip_close(ring, lane);
    };
    trip(std::integral_constant<int, 0>{});
    if constexpr (G::NH > 1) trip(std::integral_constant<int, 1>{});
    if constexpr (G::NH > 2) trip(std::integral_constant<int, 2>{});
  };
  static_for<G::NC - 1>([&](auto m_tag) { chunk(m_tag); });
  if constexpr (G::kTailOK) {
    if (ring.ktail) {
      trip_open(ring, lane);
      rings_mfma_tail<HP, MODE>(out, ring.slot(ring.par) + lane * 4, tail_to_b(in[T - 1], c, g)[0], active,
                          [&] { rings_stage(ring, wb, W, nextW, G::kTripsTail - 1, wave, lane); });
      trip_close(ring, lane);
      return;
    }
  }
  chunk(std::integral_constant<int, G::NC - 1>{});
}

// ---------------------------------------------------------------------------------------------
// One interface over both edge-GEMM engines.  Matrix offsets are the fp32 weight buffer's float offsets
// in both cases: the split image of the matrix at float offset W lives at float offset 2 W of its own buffer (a split
// matrix is 1.5x, for an odd tile count up to 1.62x, the size of the fp32 one; the holes are never touched).
// ---------------------------------------------------------------------------------------------
template <int HP, int SP>  // SP: 0 = fp32 matrix instructions, 1 / 2 = split operands with the full / half ring (SplitGeo)
struct EdgeRing {
  using type = RingS<HP, SP>;
  static constexpr int kFloats = 2 * SplitGeo<HP, SP>::kSlotFloats + (GAUDI_TRIP_FLAGS ? 4 : 0);
};
template <int HP>
struct EdgeRing<HP, 0> {
  using type = Ring<HP>;
  static constexpr int kFloats = 2 * (HP / 16) * 256;
};
__host__ __device__ constexpr int edge_ring_floats(int HP, int mode) {
  const int T = HP / 16;
  const int CH = (mode == 2 || (T + 1) / 2 == 1) && T > 1 ? (T + 1) / 2 : T;  // = SplitGeo<HP, mode>::CH
  return mode ? 2 * CH * 3 * 256 + (GAUDI_TRIP_FLAGS ? 4 : 0) : 2 * T * 256;
}
static_assert(edge_ring_floats(32, 1) == EdgeRing<32, 1>::kFloats && edge_ring_floats(208, 1) == EdgeRing<208, 1>::kFloats &&
                  edge_ring_floats(48, 2) == EdgeRing<48, 2>::kFloats && edge_ring_floats(208, 2) == EdgeRing<208, 2>::kFloats,
              "host LDS planning and SplitGeo disagree");
__device__ __forceinline__ int split_off(int W) { return W < 0 ? -1 : 2 * W; }

template <int HP>
__device__ __forceinline__ void er_init(Ring<HP>& r, float* base, bool ktail, const float*) {
  r.base = base;
  r.par = 0;
  r.ktail = ktail;
}
template <int HP, int MODE>
__device__ __forceinline__ void er_init(RingS<HP, MODE>& r, float* base, bool ktail, const float* ws) {
  r.base = base;
  r.par = 0;
  r.ktail = ktail;
#if GAUDI_TRIP_FLAGS
  // the counters live behind the two slots (EdgeRing::kFloats counts them); every phase starts from zero behind a barrier
  r.fl = (lds_int*)(base + 2 * SplitGeo<HP, MODE>::kSlotFloats);
  r.use0 = r.use1 = 0;
  r.pend = false;
  r.pend_slot = 0;
  __syncthreads();
  if (threadIdx.x < 4) r.fl[threadIdx.x] = 0;
  __syncthreads();
#endif
#if GAUDI_SPLIT_GLDS
  r.gbase = ws;
#else
  (void)ws;
#endif
}
template <int HP>
__device__ __forceinline__ void er_start(Ring<HP>& r, const WBuf& wb, int W, int wave, int lane) {
  ring_start<HP>(r, wb, W, wave, lane);
}
template <int HP, int MODE>
__device__ __forceinline__ void er_start(RingS<HP, MODE>& r, const WBuf& wb, int W, int wave, int lane) {
  rings_start(r, wb, split_off(W), wave, lane);
}
template <int HP>
__device__ __forceinline__ void er_gemm_pq(f4 (&acc)[HP / 16], Ring<HP>& ring, const WBuf& wb, int W, int nextW, const float* sB2,
                                           const float* sCr, const float* sCd, const float* pp, const float* qq, float r, float d0,
                                           bool active, int wave, int lane STAMP_DECL) {
  edge_gemm_pq<HP>(acc, ring, wb, W, nextW, sB2, sCr, sCd, pp, qq, r, d0, active, wave, lane STAMP_ARGS);
}
template <int HP, int MODE>
__device__ __forceinline__ void er_gemm_pq(f4 (&acc)[HP / 16], RingS<HP, MODE>& ring, const WBuf& wb, int W, int nextW, const float* sB2,
                                           const float* sCr, const float* sCd, const float* pp, const float* qq, float r, float d0,
                                           bool active, int wave, int lane STAMP_DECL) {
  edge_gemm_pq_s(acc, ring, wb, split_off(W), split_off(nextW), sB2, sCr, sCd, pp, qq, r, d0, active, wave, lane);
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
