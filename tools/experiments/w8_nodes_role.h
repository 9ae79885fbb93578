// w8_nodes_role.h -- node-level GEMMs on fp16 pairs with the eight waves of the workgroup split by ROLE (round 6; VERDICT r5
// item 1).  Same arithmetic, same images, same accumulation order as w8_nodes_f16.h: node_gemm_h (results are bit-identical);
// what changes is WHO moves the weights.
//
//   Y[n][o] = epi( sum_k Wa[o][k] Xa[n][k] (+ sum_k Wb[o][k] Xb[n][k]) + bias[o] )
// (edm/egnn/egnn_new.py:59-73,119-128; edm/egnn_predictor/gcl.py:240-250).
//
// node_gemm_h has every wave in every part of a GEMM at the same time (split pass | K loop: buffer_load + matrix instructions |
// epilogue), so the CU's vector-memory path idles while the waves split and store, and the waves sit in buffer_load issue while
// the matrix pipe idles: 6 343 cycles per H = 196 matrix of which the byte stream needs ~3 300.  Here:
//   * waves 4-7 are LOADERS (one per SIMD).  Loader 4 + p streams exactly the units consumer p multiplies -- output tiles p,
//     p + 4, p + 8, (p + 12), chunk-major, hi then lo, then the fp32 tail units -- by LDS-DMA (global_load_lds, no registers)
//     into a private ring of R one-KiB slots, and publishes progress in an LDS word (`full` = units landed) behind a counted
//     s_waitcnt vmcnt(F) that leaves F loads in flight.  It never executes a matrix instruction and holds no weight registers;
//   * waves 0-3 are CONSUMERS: A operands by ds_read_b128 from the ring (two tiles ahead), B operands from the split copy as
//     before, matrix instructions, fold, epilogue; a consumer tells its loader what it has taken (`freed`) one tile late, when
//     the matrix instructions of that tile have issued (their operands have arrived by then: no wait of its own).
//   The hand-off is 1 : 1 between SIMD partners -- two LDS words per pair, no workgroup barrier inside the K loop and none
//   between the matrices of a call.  The loader's vector-memory issue stalls no longer stand in front of matrix instructions,
//   and the ring runs across matrix boundaries: while the consumers fold and store matrix m and while everybody splits the
//   input of matrix m + 1, the loaders fill the ring with the first R units of m + 1.
//   * all eight waves still split the input rows (a loader's ring is full at that point anyway) and meet at the barriers.
// Barriers are matched by COUNT: a loader executes as many s_barrier as a consumer, at the points where it would block anyway.
#pragma once
#include "w8_nodes_f16.h"

namespace gaudi {
namespace w8 {

constexpr int kConsumers = 4;  // waves 0..3 multiply, wave 4 + p streams for wave p (SIMD partners: waves w and w + 4)

#ifndef GAUDI_ROLE_RING
#define GAUDI_ROLE_RING 12  // ring slots (KiB) per pair
#endif
#ifndef GAUDI_ROLE_FLIGHT
#define GAUDI_ROLE_FLIGHT 4  // loads a loader leaves in flight behind its publish point
#endif
#ifndef GAUDI_ROLE_ABLATE
#define GAUDI_ROLE_ABLATE 0  // microbenchmark only (timing, wrong results): 1 = loaders never wait for a free slot and consumers never wait for data (the
                             // two sides run free: max of the two floors); 2 = no DMA instruction (the handshake alone); 4 = the consumers skip their K
                             // loops (with 1: the loaders' floor); 8 = the loaders skip their streams (with 1: the consumers' floor)
#endif
constexpr int kRoleAblate = GAUDI_ROLE_ABLATE;

template <int HP>
struct RoleGeo {
  static constexpr int T = HP / 16, nc = nh_chunks(HP);
  static constexpr bool odd = nh_odd(HP);
  static constexpr int NTW = (T + kConsumers - 1) / kConsumers;  // tile slots per consumer (the last one may be empty)
  __host__ __device__ static constexpr int ntw(int p) { return p < T ? (T - p + kConsumers - 1) / kConsumers : 0; }
  __host__ __device__ static constexpr int pairs(int p) { return ntw(p) * (nc + (odd ? 1 : 0)); }  // slot pairs per matrix and consumer (a tail unit takes a pair)
};
// LDS of the role-split form: the four rings + the control words
__host__ __device__ constexpr int role_ring_floats(int R) { return kConsumers * R * 256 + 16; }

// Per-wave view of its pair's ring (wave-uniform values)
struct RoleState {
  float* ring;               // LDS: this pair's R slots
  volatile uint32_t* full;   // LDS word: units landed (written by the loader)
  volatile uint32_t* freed;  // LDS word: units taken (written by the consumer)
  uint32_t seq;              // units issued (loader) / taken (consumer), monotonic
  uint32_t seen;             // the other side's counter as last read
  int slot;                  // seq % R
  uint32_t pub;              // loader: units published
  int Wk, k;                 // loader: pairs 0 .. k-1 of the matrix at Wk were issued ahead of its call (Wk < 0: nothing)
};
template <int R>
__device__ __forceinline__ void role_init(RoleState& s, float* region, int wave, int tid) {
  const int p = wave & (kConsumers - 1);
  uint32_t* ctl = (uint32_t*)(region + kConsumers * R * 256);
  s.ring = region + p * R * 256;
  s.full = ctl + 2 * p;
  s.freed = ctl + 2 * p + 1;
  s.seq = s.seen = s.pub = 0;
  s.slot = 0;
  s.Wk = -1;
  s.k = 0;
  if (tid < 2 * kConsumers) ctl[tid] = 0;  // (the caller's next barrier publishes the zeros)
}
__device__ __forceinline__ uint32_t role_peek(volatile uint32_t* w) {
  // (inline assembly: the compiler's wait-count pass must not tie this LDS read to the LDS-DMA loads in flight)
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)w) : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void role_poke(volatile uint32_t* w, uint32_t v, int lane) {
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)w), "v"(v) : "memory");
}

// ---- loader side -----------------------------------------------------------------------------------------------------------
// A wave alone on its SIMD issues one instruction every ~5 cycles: the first form of this loop (per-unit bookkeeping, ~55 instructions
// per KiB) ran at 330 cycles per unit where the bare LDS-DMA loop takes 77-106 (tools/ldsdma_rate_microbench.hip).  So: units travel
// in PAIRS (a tile's hi and lo pieces: one M0, the second load through the instruction's immediate offset, which applies to the
// global and the LDS address alike), the ring has an even number of slots so that a pair never wraps, counters live in SGPRs, and every
// lane writes the progress word (no exec masking).
// one pair: wait for two free slots, two LDS-DMA loads (or ONE, narrow or wide: a tail unit, which still takes a pair of slots),
// publish what has landed behind F loads in flight
template <int R, int F, int KIND /* 0: hi + lo, 1: one 1 KiB unit, 2: one 256 B unit (4 bytes per lane) */>
__device__ __forceinline__ void role_issue2(RoleState& s, const float* gsrc, int lane) {
  static_assert(R % 2 == 0, "pairs must not wrap");
  if (!(kRoleAblate & 1)) {
    while ((int)(s.seq - s.seen) > R - 2) {
      s.seen = role_peek(s.freed);
      if ((int)(s.seq - s.seen) > R - 2) __builtin_amdgcn_s_sleep(1);
    }
  }
  constexpr int RW = (kRoleAblate & 1) && R > 12 ? 12 : R;
  float* dst = s.ring + s.slot * 256;
  if (!(kRoleAblate & 2)) {
    if (KIND == 0) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + lane * 4), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + lane * 4), (__attribute__((address_space(3))) void*)dst, 16, 1024, 0);
    } else {
      // a tail unit fills both slots of its pair (the same bytes twice): every pair is two loads, and `loads in flight` counts slots
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (KIND == 2)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + lane), (__attribute__((address_space(3))) void*)(dst + h * 256), 4, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + lane * 4), (__attribute__((address_space(3))) void*)(dst + h * 256), 16, 0,
                                           0);
      }
    }
  }
  s.seq += 2;
  s.slot = s.slot + 2 >= RW ? 0 : s.slot + 2;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F) : "memory");
  const uint32_t landed = s.seq - (uint32_t)F - (uint32_t)(F & 1);
  if ((int)(landed - s.pub) > 0) {
    s.pub = landed;
    asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)s.full), "v"(landed) : "memory");
  }
}
// everything issued has landed and is published (end of a node phase: other code may use vmcnt again)
__device__ __forceinline__ void role_flush(RoleState& s, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (s.pub != s.seq) {
    s.pub = s.seq;
    role_poke(s.full, s.pub, lane);
  }
}
// pairs [k0, k1) of the image at fp32 float offset W, for pair p: pair k = (chunk k / ntw, tile slot k % ntw), then one pair per tail tile
template <int HP, int R, int F>
__device__ __forceinline__ void role_stream(RoleState& s, const float* gimg, int W, int p, int lane, bool ktail, int k0, int k1) {
  using G = RoleGeo<HP>;
  if (kRoleAblate & 8) return;
  const int ntw = G::ntw(p), main = ntw * G::nc;
  const float* img = gimg + 2 * W + p * 512;
  int m = 0, u = k0;
  while (u >= ntw && m < G::nc) {
    u -= ntw;
    ++m;
  }
  const float* src = img + m * nh_chunk_floats(HP) + u * (kConsumers * 512);
  int k = k0;
  const int kmain = k1 < main ? k1 : main;
#pragma unroll 1
  for (; k < kmain; ++k) {
    role_issue2<R, F, 0>(s, src, lane);
    src += kConsumers * 512;
    if (++u == ntw) {
      u = 0;
      src += nh_chunk_floats(HP) - ntw * (kConsumers * 512);
    }
  }
  if constexpr (G::odd) {
    const float* ts = gimg + 2 * W + nh_tail_off(HP) + (p + kConsumers * (k - main)) * 256;
#pragma unroll 1
    for (; k < k1; ++k) {
      if (ktail) role_issue2<R, F, 2>(s, ts, lane);
      else role_issue2<R, F, 1>(s, ts, lane);
      ts += kConsumers * 256;
    }
  }
}
// The first units of the matrix that follows -- as many as the ring holds: they land while the consumers fold and store and while
// everybody splits the next input.  A matrix shorter than the flight depth (or none at all) would leave the current matrix's last
// units unpublished: flush instead.
template <int HP, int R, int F>
__device__ __forceinline__ void role_prefill(RoleState& s, const float* gimg, int nextW, int p, int lane, bool ktail) {
  const int U = RoleGeo<HP>::pairs(p);
  const int pre = nextW >= 0 ? (U < R / 2 ? U : R / 2) : 0;
  if (pre > 0) role_stream<HP, R, F>(s, gimg, nextW, p, lane, ktail, 0, pre);
  s.Wk = pre > 0 ? nextW : -1;
  s.k = pre;
  if (2 * pre < F + 2) role_flush(s, lane);
}
// The loader's share of one call: the units of Wa (and Wb) that were not issued ahead, then the head of nextW.
template <int HP, bool TWO, int R, int F>
__device__ __forceinline__ void role_loader_call(RoleState& s, const float* gimg, int Wa, int Wb, int nextW, int p, int lane, bool ktail) {
  const int U = RoleGeo<HP>::pairs(p);
  role_stream<HP, R, F>(s, gimg, Wa, p, lane, ktail, s.Wk == Wa ? s.k : 0, U);
  if constexpr (TWO) role_stream<HP, R, F>(s, gimg, Wb, p, lane, ktail, 0, U);
  role_prefill<HP, R, F>(s, gimg, nextW, p, lane, ktail);
}

// ---- consumer side ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void role_wait_full(RoleState& s, uint32_t need) {
  if (kRoleAblate & 1) return;
  while ((int)(s.seen - need) < 0) s.seen = role_peek(s.full);
}
struct RoleA {
  u4 h, l;
};
// request the two units (hi, lo) of the next tile; `has` (wave-uniform): the tile exists -- a slot without a tile reads whatever
// stands at the ring's head and multiplies it into accumulators nobody stores (the same instruction stream for every consumer)
template <int R>
__device__ __forceinline__ RoleA role_take2(RoleState& s, bool has, int lane) {
  const uint32_t inc = has ? 2u : 0u;
  role_wait_full(s, s.seq + inc);
  constexpr int RW = (kRoleAblate & 1) && R > 12 ? 12 : R;
  const float* src = s.ring + s.slot * 256 + lane * 4;
  RoleA a;
  a.h = *(const u4*)src;
  a.l = *(const u4*)(src + 256);
  s.seq += inc;
  const int s2 = s.slot + 2 >= RW ? 0 : s.slot + 2;
  s.slot = has ? s2 : s.slot;
  return a;
}

// One node GEMM of the workgroup, role-split.  Call with ALL waves; arguments as node_gemm_h.  The split copies and their barrier
// are shared by both roles; then waves >= 4 stream, waves < 4 multiply and store.  The caller places the closing barrier as before
// (an LDS barrier: s_waitcnt lgkmcnt(0) + s_barrier -- a __syncthreads would drain the loaders' DMA queue).
template <int HP, int EPI, bool TWO, int MAXNT, int R = GAUDI_ROLE_RING, int F = GAUDI_ROLE_FLIGHT, bool FL = false>
__device__ __forceinline__ void node_gemm_r(RoleState& rs, const float* gimg, int Wa, const float* sXa, bool do_split_a, int Wb, const float* sXb,
                                            const float* sBias, float* sY, const float* sRes, const float* sMask, int N, int wave, int lane,
                                            const NodeCtxH& cx, int nextW = -1, float* gPre = nullptr, uint32_t* sMaxOut = nullptr) {
  using G = RoleGeo<HP>;
  constexpr int T = G::T, LD = HP + 4, NTW = G::NTW, nc = G::nc;
  static_assert(R >= 8 && R % 2 == 0 && F >= 2 && F + 2 * RoleGeo<HP>::NTW <= R && F + 6 <= R,
                "ring depth: a consumer holds three tiles (six slots) while it waits for the next, and all its tail units (two slots each) before it hands them back");
  const int nt = (MAXNT < 2 || N <= 16) ? 1 : (MAXNT < 3 || N <= 32) ? 2 : 3;
  const bool seq = TWO && cx.split_b == cx.split_a;
  const SplitBufH sa{cx.split_a, nt, cx.scales}, sb{cx.split_b, nt, cx.scales + kScaleFloatsH};
  const int p = wave & (kConsumers - 1);
  // ---- both roles: the split copies
  if (!(kAblateH & 1)) {
    const int ls = FL ? fresh(lane) : lane;
    if (do_split_a) split_rows_h<HP>(sa, sXa, N, wave, ls);
    if (TWO && !seq) split_rows_h<HP>(sb, sXb, N, wave, ls);
    if (do_split_a || (TWO && !seq)) lds_barrier();
  }
  if (wave >= kConsumers) {
    // ---- loader
    if (TWO && seq) {  // the second input is split in the middle of the call: two more barriers, the ring full across them
      role_stream<HP, R, F>(rs, gimg, Wa, p, lane, cx.ktail, rs.Wk == Wa ? rs.k : 0, G::pairs(p));
      role_prefill<HP, R, F>(rs, gimg, Wb, p, lane, cx.ktail);
      lds_barrier();
      split_rows_h<HP>(sb, sXb, N, wave, FL ? fresh(lane) : lane);
      lds_barrier();
      role_loader_call<HP, false, R, F>(rs, gimg, Wb, -1, nextW, p, lane, cx.ktail);
    } else {
      role_loader_call<HP, TWO, R, F>(rs, gimg, Wa, Wb, nextW, p, lane, cx.ktail);
    }
    return;
  }
  // ---- consumer
  const int c = lane & 15, g = lane >> 4;
  const int bpos = nh_bpos(c, g);
  const int ntw = G::ntw(p);
  f4 acc0[MAXNT][NTW], acc1[MAXNT][NTW], y[MAXNT][NTW];
#pragma unroll
  for (int j = 0; j < MAXNT; ++j)
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int t = p + kConsumers * u < T ? p + kConsumers * u : 0;
      acc0[j][u] = splat(0.f);
      acc1[j][u] = splat(0.f);
      y[j][u] = sBias != nullptr ? *(const f4*)(sBias + 16 * t + 4 * g) : splat(0.f);
    }
  struct BH {
    u4 h, l;
  };
  auto bld = [&](const SplitBufH& s_, int i, int j) {
    const float* q = s_.chunk(i) + bpos + j * 512;
    return BH{*(const u4*)q, *(const u4*)(q + 256)};
  };
  auto mm = [&](const RoleA& a, const BH& b, int j, auto u_tag) {
    constexpr int u = decltype(u_tag)::value;
    if (kAblateH & 2) {
      asm volatile("" ::"v"(a.h), "v"(a.l), "v"(b.h), "v"(b.l));
      return;
    }
    acc1[j][u] = mfma_h(a.h, b.l, acc1[j][u]);
    acc0[j][u] = mfma_h(a.h, b.h, acc0[j][u]);
    acc1[j][u] = mfma_h(a.l, b.h, acc1[j][u]);
  };
  // tail weights of one source: k-step q of the wave's tiles, one ring unit per tile
  auto take_tail = [&](float (&tw)[NTW][4]) {
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const bool has = u < ntw;
      const uint32_t inc = has ? 2u : 0u;
      role_wait_full(rs, rs.seq + inc);
      const float* src = rs.ring + rs.slot * 256 + lane;
      tw[u][0] = src[0];
      if (!cx.ktail) {
        tw[u][1] = src[64];
        tw[u][2] = src[128];
        tw[u][3] = src[192];
      } else {
        tw[u][1] = tw[u][2] = tw[u][3] = 0.f;
      }
      rs.seq += inc;
      rs.slot = has ? (rs.slot + 2 >= R ? 0 : rs.slot + 2) : rs.slot;
    }
  };
  auto fold = [&](const SplitBufH& s_, const float (&tw)[NTW][4]) {
    static_for<MAXNT>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      if (j < nt) {
        if constexpr (G::odd) {
          const float* xt = s_.tail(HP) + j * 256 + g * 16 + c;
#pragma unroll
          for (int u = 0; u < NTW; ++u) y[j][u] = mfma1(tw[u][0], xt[0], y[j][u]);
          if (!cx.ktail) {
#pragma unroll
            for (int q = 1; q < 4; ++q)
#pragma unroll
              for (int u = 0; u < NTW; ++u) y[j][u] = mfma1(tw[u][q], xt[64 * q], y[j][u]);
          }
        }
        const float sc = s_.scale(HP)[j * 16 + c] * cx.winv;
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
          y[j][u] = y[j][u] + (acc0[j][u] + acc1[j][u] * (1.0f / kLoScale)) * sc;
          acc0[j][u] = splat(0.f);
          acc1[j][u] = splat(0.f);
        }
      }
    });
  };
  // one source: nc chunks x NTW tile slots, straight-line; A units two tiles ahead in three register sets
  auto source = [&](const SplitBufH& s_) {
    constexpr int NS = nc * NTW;  // steps
    RoleA a[3];
    uint32_t took[3] = {0u, 0u, 0u};  // units each register set holds (0: an empty tile slot)
    auto has_of = [&](int st) { return (st % NTW) < ntw; };
    uint32_t done = rs.seq;  // units handed back: lags the requests by the tiles in flight
    a[0] = role_take2<R>(rs, has_of(0), lane);
    took[0] = has_of(0) ? 2u : 0u;
    if constexpr (NS > 1) {
      a[1] = role_take2<R>(rs, has_of(1), lane);
      took[1] = has_of(1) ? 2u : 0u;
    }
    BH bcur = bld(s_, 0, 0), bnext = bcur;
    static_for<NS>([&](auto s_tag) {
      constexpr int st = decltype(s_tag)::value;
      constexpr int i = st / NTW, u = st % NTW, cur = st % 3, nx = (st + 2) % 3;
      if constexpr (u == 0 && i + 1 < nc) bnext = bld(s_, i + 1, 0);
      if constexpr (st + 2 < NS) {
        a[nx] = role_take2<R>(rs, has_of(st + 2), lane);
        took[nx] = has_of(st + 2) ? 2u : 0u;
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(a[cur], bcur, 0, std::integral_constant<int, u>{});
      if constexpr (MAXNT >= 2) {
        if (nt >= 2) {
          mm(a[cur], bld(s_, i, 1), 1, std::integral_constant<int, u>{});
          if constexpr (MAXNT >= 3) {
            if (nt >= 3) mm(a[cur], bld(s_, i, 2), 2, std::integral_constant<int, u>{});
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the units of this step are in registers (its matrix instructions have issued): hand their slots back
      done += took[cur];
      if (took[cur] != 0u && !(kRoleAblate & 1))
        asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)rs.freed), "v"(done) : "memory");
      if constexpr (u == NTW - 1) bcur = bnext;
    });
  };
  float ta[NTW][4], tb[NTW][4];
  if (!(kRoleAblate & 4)) source(sa);
  if constexpr (G::odd) take_tail(ta);
  fold(sa, ta);
  if (G::odd && !(kRoleAblate & 1)) role_poke(rs.freed, rs.seq, lane);  // (the tail units were consumed by fold's matrix instructions)
  if constexpr (TWO) {
    if (seq) {
      lds_barrier();
      split_rows_h<HP>(sb, sXb, N, wave, FL ? fresh(lane) : lane);
      lds_barrier();
    }
    if (!(kRoleAblate & 4)) source(sb);
    if constexpr (G::odd) take_tail(tb);
    fold(sb, tb);
    if (G::odd && !(kRoleAblate & 1)) role_poke(rs.freed, rs.seq, lane);
  }
  const int le = FL ? fresh(lane) : lane, ce = le & 15, ge = le >> 4;
  static_for<MAXNT>([&](auto j_tag) {
    constexpr int j = decltype(j_tag)::value;
    if (j < nt) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        const int t = p + kConsumers * u;
        const int nd = j * 16 + ce;
        f4 yy = y[j][u];
        if (t < T && nd < N) {
          float* dst = sY + nd * LD + 16 * t + 4 * ge;
          const bool pad = G::odd && cx.ktail && t == T - 1 && ge > 0;
          if (pad) yy = splat(0.f);
          if (gPre != nullptr) nstash_store((f4*)(gPre + nd * HP + 16 * t + 4 * ge), yy);
          if (sMaxOut != nullptr) atomicMax(sMaxOut + nd, umax(umax(absbits(yy[0]), absbits(yy[1])), umax(absbits(yy[2]), absbits(yy[3]))));
          if (!pad) {
            if (EPI == EPI_SILU) yy = silu4(yy);
            if (EPI == EPI_RESIDUAL_MASK) {
              const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * ge);
              yy = (r + yy) * sMask[nd];
            }
            if (EPI == EPI_MUL_DSILU) {
              const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * ge);
              yy = (f4){yy[0] * dsilu_f(r[0]), yy[1] * dsilu_f(r[1]), yy[2] * dsilu_f(r[2]), yy[3] * dsilu_f(r[3])};
            }
            if (EPI == EPI_ACCUM) yy = *(const f4*)(sRes + nd * LD + 16 * t + 4 * ge) + yy;
          }
          *(f4*)dst = yy;
        }
      }
    }
  });
}

}  // namespace w8
}  // namespace gaudi
