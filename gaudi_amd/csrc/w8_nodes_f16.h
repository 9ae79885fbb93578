// w8_nodes_f16.h -- node-level GEMMs of the 8-wave kernels on the fp16 matrix pipe with fp32-equivalent accuracy, at the SAME
// bytes per weight as the fp32 form (round 5).
//
//   Y[n][o] = epi( sum_k Wa[o][k] Xa[n][k] (+ sum_k Wb[o][k] Xb[n][k]) + bias[o] )
// for the node columns of a workgroup: the P / Q / node-MLP GEMMs of every layer and their transposes in the reverse pass
// (edm/egnn/egnn_new.py:59-73,119-128; edm/egnn_predictor/gcl.py:240-250) -- 183 matrices per guided step, each streamed from L2
// once per workgroup.  The fp32 form (w8_common.h: node_gemm) runs them on v_mfma_f32_16x16x4_f32, which delivers the VECTOR
// fp32 rate (64 FLOP/clk/SIMD): 5 408 of its 8 355 cycles per H = 196 matrix are matrix-instruction time.
//
// Arithmetic.  An fp32 number x with |x| < 2^15 is written x = hi + lo * 2^-11 with hi = fp16(x) and lo = fp16((x - hi) * 2^11)
// (round to nearest): hi carries 11 significant bits, the remainder x - hi is EXACT in fp32 and lo carries its leading 11, so
// |x - hi - lo 2^-11| <= 2^-22 |x| -- and because lo has its own exponent the absolute floor is 2^-36 (fp16's subnormal spacing
// / 2^11), 50 binades below the largest representable value: small entries beside large ones lose nothing that matters.
// A product is accumulated in fp32 from three piece products on v_mfma_f32_16x16x32_f16 (16x the fp32 instruction's rate):
//     acc0 += w_hi x_hi          acc1 += w_hi x_lo + w_lo x_hi          y = (acc0 + 2^-11 acc1) * 2^-(s_w + s_x)
// (dropped: w_lo x_lo 2^-22).  Piece products are exact in fp32 (11 x 11 bits); the error beside the fp32 accumulation is
// <= 3 * 2^-22 |w x| per term, the same order as the fp32 instruction's own rounding chain over K = 196 terms (measured against
// float64 in tools/node_gemm_h_microbench.hip and tests/test_gpu_round5.py: the same level as v_mfma_f32_16x16x4_f32).
// Range.  fp16 overflows at 65 504, so both operands are scaled by powers of two (exact): the weights once on the host with ONE
// exponent per network (gaudi_hip.hip: pack_matrix_f16; max |w| of all node matrices -> [2^13, 2^14)), the activations per
// NODE and per GEMM input on the device (the row's max exponent -> 2^14: split_rows_h), undone in the epilogue.  NaN
// propagates as in fp32; an infinite activation gives NaN where fp32 gives +-inf (hi = inf, x - hi = NaN) -- the sampler
// scrubs both the same way (models.py:138-141).  The host refuses the form (-> fp32 node GEMMs, with a warning: gaudi_last_warning)
// for weight sets with infinities or a matrix whose largest entry lies more than 2^12 below the network's (gaudi_hip.hip: NodeScale).
//
// Bytes.  hi and lo are 2 + 2 bytes: a matrix image is exactly the fp32 matrix's size -- units of 1 KiB ordered
// [K chunk of 32][output tile][piece], lane L = (row L & 15, inputs 8 (L >> 4) .. +7 of the chunk).  An odd tile count (208 =
// 13 tiles, 48 = 3) leaves a last half chunk: those 16 inputs stay fp32, unscaled -- a trailing block [tile][k-step q][lane
// (row, g)] = W[row][16 (T-1) + 4 q + g], run as v_mfma_f32_16x16x4_f32 steps: ONE step for H % 16 == 4 (196: four valid
// inputs; only that step's 256 B per tile are ever loaded), four otherwise.  Images live in the split buffer at float offset
// 2 W (W = the fp32 buffer's offset), in the slots the node matrices leave empty there.
//
// Schedule.  Output tile t -> wave t & 7 (as the fp32 form).  The activations of a GEMM input are split ONCE by all waves into
// LDS (wave w: rows w, w + 8, ...; one DPP max-scan per row, no cross-wave step) in B-operand order -- two conflict-free
// ds_read_b128 per chunk and column tile; the weight stream runs kDepthH chunks ahead in registers, across the two sources of
// a GEMM and across calls (the next matrix's first chunks travel while this one drains).  ONE body serves every wave and every
// column-tile count: the registers of the weight stream are defined at one program point per chunk (a wave without a second tile
// loads with out-of-range lanes, which fetch nothing); with one body per case hipcc joined the cases through register copies of
// loads in flight, i.e. a wait for the whole stream at every call.  Accumulation order per output element: K chunks in order,
// per chunk w_hi x_lo, w_hi x_hi, w_lo x_hi; then the fp32 tail steps; sources in order -- independent of the column-tile count
// and of the wave's tile count (packed launches stay bit-identical to unpacked ones).
#pragma once
#include "w8_split.h"

namespace gaudi {
namespace w8 {


constexpr float kLoScale = 2048.f;  // 2^11
#ifndef GAUDI_NODE_ABLATE
#define GAUDI_NODE_ABLATE 0  // microbenchmark only (timing, wrong results): 1 = no split pass, 2 = no matrix instructions, 4 = no weight loads
#endif
constexpr int kAblateH = GAUDI_NODE_ABLATE;
// microbenchmark only (-DGAUDI_NODE_STAMPS=1): cycle sums of the parts of a node GEMM, per wave (s_memtime waits for lgkmcnt(0): shares, not
// timings -- DESIGN section 7, lesson 5)
#ifndef GAUDI_NODE_STAMPS
#define GAUDI_NODE_STAMPS 0
#endif
#ifndef GAUDI_NODE_PRIO
#define GAUDI_NODE_PRIO 1  // 1 = waves 4-7 at s_setprio 1 inside the K loop (kernels with one column tile; round 6), 2 = waves 0-3, 0 = off
#endif
struct NodeStampH {
  unsigned long long sum[8], last;
  __device__ __forceinline__ void start() { last = __builtin_amdgcn_s_memtime(); }
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    sum[i] += t - last;
    last = t;
  }
};
#if GAUDI_NODE_STAMPS
#define NSTAMP(i) do { if (ns != nullptr) ns->mark(i); } while (0)
#else
#define NSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f4 mfma_h(const u4 a, const u4 b, const f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u4 ldu4h(const WBuf& wh, int off_floats, int lane) {
  return __builtin_bit_cast(u4, ldw4n(wh, off_floats, lane));
}

// ---- geometry shared with the host packer (gaudi_hip.hip: pack_matrix_f16)
__host__ __device__ constexpr bool nh_odd(int HP) { return ((HP / 16) & 1) != 0; }          // a last half chunk: the fp32 tail block
__host__ __device__ constexpr int nh_chunks(int HP) { return HP / 32; }                      // fp16 chunks of 32 inputs
__host__ __device__ constexpr int nh_chunk_floats(int HP) { return (HP / 16) * 2 * 256; }    // one K chunk of an image
__host__ __device__ constexpr int nh_tail_off(int HP) { return nh_chunks(HP) * nh_chunk_floats(HP); }  // [tile][q][64] floats
// k-steps of the tail block that hold weights: one when only four inputs of the last tile exist (H % 16 == 4), else four
__host__ __device__ constexpr int nh_tail_steps(bool ktail) { return ktail ? 1 : 4; }
// split copy of one GEMM input in LDS, NCT column tiles of 16 nodes: [chunk][column tile][piece][64 x 16 B] -- with one column tile
// 64 B of padding per chunk (the splitting wave's 8-byte stores of one row then fall on distinct banks; with more tiles the copy
// must fit the half ring exactly: 2 x 6 chunks x 2 KiB at H = 192) -- then the tail's fp32 inputs [column tile][node][16].  The
// per-node descale factors live in a small array of their own (kScaleFloatsH per input).
__host__ __device__ constexpr int nh_chunk_stride(int nct) { return nct * 512 + (nct == 1 ? 16 : 0); }
__host__ __device__ constexpr int nh_split_floats(int HP, int nct) {
  return nh_chunks(HP) * nh_chunk_stride(nct) + (nh_odd(HP) ? nct * 256 : 0);
}
constexpr int kScaleFloatsH = 48;  // [column tile <= 3][16]

struct SplitBufH {
  float* base;
  int nct;    // column tiles
  float* sc;  // LDS [kScaleFloatsH]: the rows' descale factors
  __device__ __forceinline__ float* chunk(int m) const { return base + m * nh_chunk_stride(nct); }
  __device__ __forceinline__ float* tail(int HP) const { return base + nh_chunks(HP) * nh_chunk_stride(nct); }
  __device__ __forceinline__ float* scale(int) const { return sc; }
};

// max over the wave of an unsigned value that every lane holds (DPP scan inside the 16-lane rows, then the four row ends)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  auto shr = [](uint32_t x, auto sh_tag) {
    constexpr int SH = decltype(sh_tag)::value;
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + SH, 0xF, 0xF, true);
  };
  v = umax(v, shr(v, std::integral_constant<int, 1>{}));
  v = umax(v, shr(v, std::integral_constant<int, 2>{}));
  v = umax(v, shr(v, std::integral_constant<int, 4>{}));
  v = umax(v, shr(v, std::integral_constant<int, 8>{}));
  const uint32_t a = __builtin_amdgcn_readlane((int)v, 15), b = __builtin_amdgcn_readlane((int)v, 31),
                 c = __builtin_amdgcn_readlane((int)v, 47), d = __builtin_amdgcn_readlane((int)v, 63);
  return umax(umax(a, b), umax(c, d));
}


// all waves: rows [0, N) of X ([N][HP+4], LDS -- or global for the kernels with node buffers in global memory) -> split
// B operands.  Wave w owns rows w, w + 8, ...: lane j holds inputs 4 j .. 4 j + 3 of the row; two rows travel together so that
// their LDS round trips and DPP wait states cover each other.  The caller places a barrier between this and the GEMM's reads.
// Columns >= N of the last tile are left as they are: a matrix-instruction column depends on its own B column only and those
// results are never stored.
// Where column c's inputs 8 g .. 8 g + 7 of a chunk (16 bytes of fp16) sit inside the 1 KiB B unit, in floats.  The readers are the
// matrix instructions' B operands: lane (c, g) of every wave, 16 bytes each, once per chunk and column tile -- 16 consecutive lanes
// (one g) must cover 256 consecutive bytes.  The writers are the split pass's rows: one column c per instruction, lanes (chunk, g,
// half) -- the XOR spreads the four g of a column over the four quads of a 16-bank block, the chunk stride's padding spreads the
// chunks over the four blocks.  (Through the first version of round 5 the unit was column-major, 4 c + g: a row's store was one
// contiguous 64 bytes, but the reads of 16 lanes were 64 bytes apart -- four lanes per bank: 0.48 conflict cycles per LDS-active
// cycle in the GEMM without its split pass, profiles/r05e_lds_conflicts_by_gemm.txt.)
__host__ __device__ constexpr int nh_bpos(int c, int g) { return (16 * g + (c ^ g)) * 4; }
struct SplitRowH {
  f4 x;
  uint32_t mx;
};
template <int HP>
__device__ __forceinline__ SplitRowH split_row_load(const float* X, int n, int lane) {
  constexpr int LD = HP + 4;
  SplitRowH r;
  r.x = lane < HP / 4 ? *(const f4*)(X + n * LD + 4 * lane) : splat(0.f);
  // (absbits takes the element BY VALUE: __builtin_bit_cast applied to an ext-vector element expression reads element 0
  // whatever the index -- hipcc 7.2; found as rows whose largest entry was not in a lane's first slot getting the wrong scale)
  r.mx = umax(umax(absbits(r.x[0]), absbits(r.x[1])), umax(absbits(r.x[2]), absbits(r.x[3])));
  return r;
}
template <int HP>
__device__ __forceinline__ void split_row_store(const SplitBufH& sb, const f4 x, uint32_t mx, int n, int lane) {
  constexpr int T = HP / 16, nc = nh_chunks(HP);
  const int m = lane >> 3, g = (lane >> 1) & 3, half = lane & 1;
  // the row's largest exponent -> 14 (values below 2^15 < 65 504); exponents clamped so that both factors are normal numbers
  int k = 141 - (int)(mx >> 23);
  k = k > 126 ? 126 : k;
  const float s = __builtin_bit_cast(float, (uint32_t)(k + 127) << 23), inv = __builtin_bit_cast(float, (uint32_t)(127 - k) << 23);
  const float y0 = x[0] * s, y1 = x[1] * s, y2 = x[2] * s, y3 = x[3] * s;
  const uint32_t h01 = pk_f16(y0, y1), h23 = pk_f16(y2, y3);
  const f2 f01 = unpk_f16(h01), f23 = unpk_f16(h23);
  const uint32_t l01 = pk_f16((y0 - f01[0]) * kLoScale, (y1 - f01[1]) * kLoScale),
                 l23 = pk_f16((y2 - f23[0]) * kLoScale, (y3 - f23[1]) * kLoScale);
  const int ct = n >> 4, c = n & 15;
  if (m < nc) {
    float* d = sb.chunk(m) + ct * 512 + nh_bpos(c, g) + half * 2;
    *(u2*)d = (u2){h01, h23};
    *(u2*)(d + 256) = (u2){l01, l23};
  } else if (nh_odd(HP) && lane < HP / 4) {  // inputs 16 (T-1) .. +15, unscaled: the fp32 tail steps
    float* d = sb.tail(HP) + ct * 256 + 4 * (lane - 4 * (T - 1)) * 16 + c;  // [input k][column c]: the readers' 64 lanes hit 64 banks
    d[0] = x[0];
    d[16] = x[1];
    d[32] = x[2];
    d[48] = x[3];
  }
  if (lane == 0) sb.scale(HP)[ct * 16 + c] = inv;
}
template <int HP>
__device__ __forceinline__ void split_rows_h(const SplitBufH& sb, const float* X, int N, int wave, int lane) {
  for (int n = wave; n < N; n += 2 * kWaves) {
    const int n2 = n + kWaves < N ? n + kWaves : n;  // (a second copy of the same row when there is no partner: same stores)
    SplitRowH a = split_row_load<HP>(X, n, lane), b = split_row_load<HP>(X, n2, lane);
    a.mx = wave_max_u32(a.mx);
    b.mx = wave_max_u32(b.mx);
    split_row_store<HP>(sb, a.x, a.mx, n, lane);
    if (n2 != n) split_row_store<HP>(sb, b.x, b.mx, n2, lane);
  }
}

// LDS writes of every wave visible to every wave; global loads in flight stay in flight (__syncthreads would wait for them)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Weight chunks in flight per wave.  A buffer_load occupies its wave until the CU's vector-memory pipe accepts it (~50 B/clk per
// CU: the K loop of a matrix cannot be shorter than its bytes take), so a deeper queue buys nothing once the loads of one step
// are covered (depth 3, 4 and 6 measure the same: profiles/r05d_node_gemm_h_microbench.txt); what pays is issuing them in
// straight-line code -- hipcc then counts the loads in flight exactly, where any branch around a load made it wait for
// vmcnt(0) at every step.  A matrix's chunks are consumed in turns of D sets: chunk i lives in set i % D of every matrix.
#ifndef GAUDI_NODE_DEPTH
#define GAUDI_NODE_DEPTH 3
#endif
// Of the next matrix's chunks that a call loads ahead, the last GAUDI_NODE_LATE are issued AFTER its K loop -- one behind the
// loop, one behind the fold -- instead of inside it: the K loop is bound by the CU's vector-memory pipe (its waves sit in
// buffer_load issue), the split pass and the epilogue leave that pipe idle, so loads moved there cost nothing.
#ifndef GAUDI_NODE_LATE
#define GAUDI_NODE_LATE 0
#endif
template <int HP>
struct NodeGeoH {
  static constexpr int T = HP / 16;
  static constexpr int NTW = T > kWaves ? 2 : 1;  // output tiles per wave (tile t -> wave t & 7)
  static constexpr int nc = nh_chunks(HP);
  static constexpr int D = nc < GAUDI_NODE_DEPTH ? (nc > 0 ? nc : 1) : GAUDI_NODE_DEPTH;
  static constexpr bool odd = nh_odd(HP);
};
// one chunk of the wave's (up to two) output tiles: [tile][piece]
template <int NTW>
struct NodeSetH {
  u4 p[NTW][2];
};
template <int HP>
struct NodePFH {
  NodeSetH<NodeGeoH<HP>::NTW> s[NodeGeoH<HP>::D];  // chunks 0 .. D-1 of the next node GEMM, loaded ahead of the call
};

// the lane offsets of the wave's tiles: a tile the wave does not have is "loaded" with an out-of-range lane (returns 0, fetches
// nothing) -- every wave runs the same loads, none sits behind a branch
template <int HP>
struct TileLanesH {
  int l[NodeGeoH<HP>::NTW];
  __device__ __forceinline__ TileLanesH(int wave, int lane, bool on = true) {
#pragma unroll
    for (int u = 0; u < NodeGeoH<HP>::NTW; ++u) l[u] = (on && wave + kWaves * u < NodeGeoH<HP>::T) ? lane : kOOBLane;
  }
};
template <int HP>
__device__ __forceinline__ void nh_load(NodeSetH<NodeGeoH<HP>::NTW>& s, const WBuf& wh, int chunk_off, int wave, const TileLanesH<HP>& tl) {
#pragma unroll
  for (int u = 0; u < NodeGeoH<HP>::NTW; ++u) {
    const int t = wave + kWaves * u < NodeGeoH<HP>::T ? wave + kWaves * u : 0;
    s.p[u][0] = ldu4h(wh, chunk_off + (t * 2 + 0) * 256, tl.l[u]);
    s.p[u][1] = ldu4h(wh, chunk_off + (t * 2 + 1) * 256, tl.l[u]);
  }
}
// How many chunks travel ahead of a call: kAheadAll = as many as the depth (where the next GEMM follows directly), kAheadOne =
// ONE chunk (8-16 registers: across the register-tight edge phases; the call issues the others itself, and they fly while its
// input is split).
constexpr int kAheadOne = 1, kAheadAll = 99;
template <int HP, int AHEAD = kAheadAll>
__device__ __forceinline__ void node_prefetch_h(NodePFH<HP>& pf, const WBuf& wh, int W /* fp32 float offset */, int wave, int lane) {
  constexpr int nd = AHEAD < NodeGeoH<HP>::D ? AHEAD : NodeGeoH<HP>::D;
  const TileLanesH<HP> tl(wave, lane);
  static_for<nd>([&](auto d_tag) {
    constexpr int d = decltype(d_tag)::value;
    nh_load<HP>(pf.s[d], wh, 2 * W + d * nh_chunk_floats(HP), wave, tl);
  });
}

// what a call needs beside the fp32 form's arguments
struct NodeCtxH {
  float winv;       // 2^-s_w: descale of the network's node matrices
  float* split_a;   // LDS: split copy of the first source (nh_split_floats(HP, nct) floats)
  float* split_b;   // ... of the second source; == split_a: the region holds ONE input, the sources are split in turn
  bool ktail;       // H % 16 == 4: the tail block holds one k-step
  float* scales;    // LDS [2][kScaleFloatsH]: the per-node descale factors of the two sources
  float* scales_b = nullptr;  // the second source's, when they do not follow the first's (a kept copy of h: node_ctx_keep)
};

// One node GEMM of the workgroup (all waves call it; N <= 16 MAXNT node columns).
//   TWO: a second source (Wb, sXb).  do_split_a = false: an earlier call of this phase left the first source's split copy in
//   cx.split_a (P and Q share h, the two transposed GEMMs of dnpre share it).  PIN: chunks of Wa that were loaded ahead of the call
//   (pf.s[0 .. min(PIN, D))); POUT: chunks of nextW this call loads ahead.
// The caller guarantees that nobody still reads the split regions when the call starts (a barrier since their last use) and
// places a barrier between this call's stores to sY and their readers, as for the fp32 form.
template <int HP, int EPI, bool TWO, int MAXNT, int PIN = kAheadOne, int POUT = kAheadOne, bool FL = false>
__device__ __forceinline__ void node_gemm_h(const WBuf& wh, int Wa, const float* sXa, bool do_split_a, int Wb, const float* sXb,
                                            const float* sBias, float* sY, const float* sRes, const float* sMask, int N, int wave,
                                            int lane, const NodeCtxH& cx, NodePFH<HP>& pf, int nextW = -1, float* gPre = nullptr,
                                            uint32_t* sMaxOut = nullptr /* LDS [N], zeroed: max |y| bits of every node's row */,
                                            NodeStampH* ns = nullptr) {
  (void)ns;
  using G = NodeGeoH<HP>;
  constexpr int T = G::T, LD = HP + 4, NTW = G::NTW, nc = G::nc, D = G::D;
  constexpr int kIn = PIN < D ? PIN : D, kOut = POUT < D ? POUT : D;
  constexpr int kLate = nc < D ? 0 : (GAUDI_NODE_LATE < kOut ? GAUDI_NODE_LATE : kOut);  // (matrices of fewer chunks than sets: all in the loop)
  const int c = lane & 15, g = lane >> 4;
  // column tiles of this call (wave-uniform; the code below branches on it around matrix instructions and LDS traffic only)
  const int nt = (MAXNT < 2 || N <= 16) ? 1 : (MAXNT < 3 || N <= 32) ? 2 : 3;
  const bool seq = TWO && cx.split_b == cx.split_a;
  const SplitBufH sa{cx.split_a, nt, cx.scales}, sb{cx.split_b, nt, cx.scales_b != nullptr ? cx.scales_b : cx.scales + kScaleFloatsH};
  const int bpos = nh_bpos(c, g);  // the lane's float offset inside a 1 KiB B unit
  const TileLanesH<HP> tl(wave, lane);
  // a matrix that is not there (no next GEMM) is "loaded" with out-of-range lanes too
  const TileLanesH<HP> tl_next(wave, lane, nextW >= 0);
  const int Wn = nextW >= 0 ? nextW : Wa;
  // chunks that did not travel ahead of the call
  static_for<D>([&](auto d_tag) {
    constexpr int d = decltype(d_tag)::value;
    if constexpr (d >= kIn) nh_load<HP>(pf.s[d], wh, 2 * Wa + d * nh_chunk_floats(HP), wave, tl);
  });
  // tail weights (odd tile counts): k-step q of the wave's tiles; steps past the first are loaded only when they hold weights
  float ta[NTW][4], tb[NTW][4];
  if constexpr (G::odd) {
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int t = wave + kWaves * u < T ? wave + kWaves * u : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ln = (q == 0 || !cx.ktail) ? tl.l[u] : kOOBLane;
        ta[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wh.r, ln * 4, (2 * Wa + nh_tail_off(HP) + t * 256 + q * 64) * 4, 0));
        if constexpr (TWO)
          tb[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wh.r, ln * 4, (2 * Wb + nh_tail_off(HP) + t * 256 + q * 64) * 4, 0));
      }
    }
  }
  // FL: the split passes and the epilogue take their addresses from fresh(lane) (device_common.h): recomputed per call -- some
  // forty integer instructions -- instead of being hoisted out of the layer loop and spilled; the K loop's offsets stay hoisted.
  NSTAMP(0);
  if (!(kAblateH & 1)) {
    const int ls = FL ? fresh(lane) : lane;
    if (do_split_a) split_rows_h<HP>(sa, sXa, N, wave, ls);
    if (TWO && !seq) split_rows_h<HP>(sb, sXb, N, wave, ls);
    NSTAMP(1);
    if (do_split_a || (TWO && !seq)) lds_barrier();
    NSTAMP(2);
  }

  f4 acc0[MAXNT][NTW], acc1[MAXNT][NTW], y[MAXNT][NTW];
#pragma unroll
  for (int j = 0; j < MAXNT; ++j)
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int t = wave + kWaves * u < T ? wave + kWaves * u : 0;
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
  // the three piece products of one column tile against one chunk
  auto mm = [&](const NodeSetH<NTW>& s, const BH& b, int j) {
    if (kAblateH & 2) {  // (keep the operands alive)
      asm volatile("" ::"v"(s.p[0][0]), "v"(s.p[0][1]), "v"(s.p[NTW - 1][0]), "v"(s.p[NTW - 1][1]), "v"(b.h), "v"(b.l));
      return;
    }
#pragma unroll
    for (int u = 0; u < NTW; ++u) acc1[j][u] = mfma_h(s.p[u][0], b.l, acc1[j][u]);
#pragma unroll
    for (int u = 0; u < NTW; ++u) acc0[j][u] = mfma_h(s.p[u][0], b.h, acc0[j][u]);
#pragma unroll
    for (int u = 0; u < NTW; ++u) acc1[j][u] = mfma_h(s.p[u][1], b.h, acc1[j][u]);
  };
  // descale and fold the accumulators of one source into y
  auto fold = [&](const SplitBufH& s_, const float (&tw)[NTW][4]) {
    static_for<MAXNT>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      if (j < nt) {
        if constexpr (G::odd) {  // the tail's fp32 steps: inputs 16 (T-1) + 4 q + g on lane group g, unscaled operands
          const float* xt = s_.tail(HP) + j * 256 + g * 16 + c;  // [input k = g + 4 q][column c]
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
  // one source: its nc chunks in order, straight-line.  After chunk i the set is refilled with what it holds next: chunk i + D
  // of the same matrix, else chunk i % D of the matrix that follows (the second source; after the last source the next GEMM's
  // matrix, as far as POUT allows).
  auto source = [&](auto src_tag, const SplitBufH& s_) {
    constexpr int src = decltype(src_tag)::value;
    constexpr bool last_src = !TWO || src == 1;
    const int Wcur = src == 0 ? Wa : Wb;
    BH bcur = bld(s_, 0, 0);
    static_for<nc>([&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value;
      constexpr int d = i % D;
      BH bnext = bcur;
      if constexpr (i + 1 < nc) bnext = bld(s_, i + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(pf.s[d], bcur, 0);
      if constexpr (MAXNT >= 2) {
        if (nt >= 2) {  // (matrix instructions and LDS reads only: nothing of the weight stream sits behind this branch)
          mm(pf.s[d], bld(s_, i, 1), 1);
          if constexpr (MAXNT >= 3) {
            if (nt >= 3) mm(pf.s[d], bld(s_, i, 2), 2);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(kAblateH & 4)) {
        if constexpr (i + D < nc) nh_load<HP>(pf.s[d], wh, 2 * Wcur + (i + D) * nh_chunk_floats(HP), wave, tl);
        else if constexpr (!last_src) nh_load<HP>(pf.s[d], wh, 2 * Wb + d * nh_chunk_floats(HP), wave, tl);
        else if constexpr (d < kOut - kLate) nh_load<HP>(pf.s[d], wh, 2 * Wn + d * nh_chunk_floats(HP), wave, tl_next);
      }
      __builtin_amdgcn_sched_barrier(0);
      bcur = bnext;
    });
  };
  // the next matrix's chunk d, issued behind the K loop (its set's last use lies behind: chunk nc - D + d .. of the last source)
  auto late = [&](auto d_tag) {
    constexpr int d = decltype(d_tag)::value;
    if constexpr (d >= kOut - kLate && d < kOut) {
      __builtin_amdgcn_sched_barrier(0);
      if (!(kAblateH & 4)) nh_load<HP>(pf.s[d], wh, 2 * Wn + d * nh_chunk_floats(HP), wave, tl_next);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  NSTAMP(3);
#if GAUDI_NODE_PRIO
  // Round 6: the second-dispatched half of the workgroup (waves 4-7) loses every issue arbitration by age -- its K loops end last
  // and everybody waits for them at the closing barrier (MI355X_MICROARCH.md, two waves per SIMD, item 4): one priority step for it
  // while it streams.  Microbenchmark -0.3 ... -1.9 % per GEMM at one column tile (profiles/r06b_node_gemm_priority.txt); product C3
  // 206.3 -> 208.1 mol/s.  With TWO column tiles the microbenchmark gains more (-3.7 %) and the product LOSES (C4 245.0 -> 242.2,
  // pairs -0.2 %): only the kernels that run one column tile (not FL) take the step.  Priority for waves 0-3 instead: nothing.
  if constexpr (!FL)
    if ((GAUDI_NODE_PRIO == 1 && wave >= kWaves / 2) || (GAUDI_NODE_PRIO == 2 && wave < kWaves / 2)) __builtin_amdgcn_s_setprio(1);
#endif
  source(std::integral_constant<int, 0>{}, sa);
  NSTAMP(4);
  if constexpr (!TWO) late(std::integral_constant<int, kOut - kLate>{});
  fold(sa, ta);
  NSTAMP(5);
  if constexpr (TWO) {
    if (seq) {
      lds_barrier();  // (every wave is done with the first source's copy; the weight loads in flight stay in flight)
      split_rows_h<HP>(sb, sXb, N, wave, FL ? fresh(lane) : lane);
      lds_barrier();
    }
    source(std::integral_constant<int, 1>{}, sb);
    late(std::integral_constant<int, kOut - kLate>{});
    fold(sb, tb);
  }
  static_for<D>([&](auto d_tag) {
    if constexpr (decltype(d_tag)::value > kOut - kLate) late(d_tag);
  });
#if GAUDI_NODE_PRIO
  if constexpr (!FL) __builtin_amdgcn_s_setprio(0);
#endif
  const int le = FL ? fresh(lane) : lane, ce = le & 15, ge = le >> 4;
  static_for<MAXNT>([&](auto j_tag) {
    constexpr int j = decltype(j_tag)::value;
    if (j < nt) {
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        const int t = wave + kWaves * u;
        const int nd = j * 16 + ce;
        f4 yy = y[j][u];
        if (t < T && nd < N) {
          float* dst = sY + nd * LD + 16 * t + 4 * ge;
          const bool pad = G::odd && cx.ktail && t == T - 1 && ge > 0;  // rows 4 .. 15 of the last tile of an H % 16 == 4 width
          if (pad) yy = splat(0.f);
          if (gPre != nullptr) nstash_store((f4*)(gPre + nd * HP + 16 * t + 4 * ge), yy);  // stash: write once, read once
          // the row maxima the edge GEMMs' column scales are bounded with (w8_split.h): one LDS atomic per lane and tile
          if (sMaxOut != nullptr) atomicMax(sMaxOut + nd, umax(umax(absbits(yy[0]), absbits(yy[1])), umax(absbits(yy[2]), absbits(yy[3]))));
          if (!pad) {
            if (EPI == EPI_SILU) yy = silu4v(yy);
            if (EPI == EPI_RESIDUAL_MASK) {
              const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * ge);
              yy = (r + yy) * sMask[nd];
            }
            if (EPI == EPI_MUL_DSILU) {  // y * silu'(pre-activation stored in sRes); in place is safe
              const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * ge);
              yy = yy * dsilu4v(r);
            }
            if (EPI == EPI_ACCUM) yy = *(const f4*)(sRes + nd * LD + 16 * t + 4 * ge) + yy;
          }
          *(f4*)dst = yy;
        }
      }
    }
  });
  NSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// One call-site form over both node-GEMM engines.  NH = the fp16-pair form: every kernel whose edge GEMMs run on split operands
// (SP != 0) unless the build switches it off; the fp32-instruction kernels (SP = 0) keep the fp32 form and are what the host
// falls back to for weight sets the fp16 images refuse.
// ---------------------------------------------------------------------------------------------
#ifndef GAUDI_NODE_F16
#define GAUDI_NODE_F16 1
#endif
template <int SP>
struct NodeMath {
  static constexpr bool kF16 = SP != 0 && GAUDI_NODE_F16 != 0;
};
template <int HP, bool NH>
struct NodePFSel {
  using type = NodePF<HP>;
};
template <int HP>
struct NodePFSel<HP, true> {
  using type = NodePFH<HP>;
};
template <int HP, bool NH, int AHEAD, class PF>
__device__ __forceinline__ void node_prefetch_x(PF& pf, const WBuf& wb, const WBuf& wbe, int W, int N, int wave, int lane, bool tw) {
  if constexpr (NH) node_prefetch_h<HP, AHEAD>(pf, wbe, W, wave, lane);
  else node_prefetch<HP>(pf, wb, W, wave, lane, tw);
}
// Xa / Xb: the input rows where they live (LDS, or global memory for the GN kernels); XaS / XbS: their staged copies in the idle
// weight ring, which only the fp32 form of a GN kernel reads (w8_common.h: stage_rows) -- the fp16 form splits the rows straight
// from where they are.  split_a = false: the previous call's split copy of Xa is still in place.
// FL (node_gemm_h): without it the kernels keep some thirty lane-dependent LDS addresses of the split passes and epilogues alive
// across every GEMM of a layer loop -- in scratch, read back behind `s_waitcnt vmcnt(0)` one after the other.  Measured on the
// resident full-ring kernel (same code, same launch, A/B of two libraries): C4 (20 node slots, two column tiles) +2.4 %, batches of
// 1024 in pairs +2.2 %, but C3 (11 nodes, one column tile: half the epilogue work per reload) -0.4 % -- the recomputation costs
// what the reloads did.  So FL is on in the kernels that take large molecules (MR, GN, and the FR instantiation of the resident
// kernel that the host picks for more than 16 node slots) and off in the one C2 / C3 run on.
template <int HP, int EPI, bool TWO, int GN, bool NH, int PIN, int POUT, bool FL = false, class PF>
__device__ __forceinline__ void node_gemm_x(const WBuf& wb, const WBuf& wbe, int Wa, const float* Xa, const float* XaS, bool split_a, int Wb,
                                            const float* Xb, const float* XbS, const float* sBias, float* sY, const float* sRes,
                                            const float* sMask, int N, int wave, int lane, bool tw, const NodeCtxH& cx, PF& pf,
                                            int nextW = -1, float* gPre = nullptr, uint32_t* sMaxOut = nullptr) {
  if constexpr (NH) {
    node_gemm_h<HP, EPI, TWO, GN ? 3 : 2, PIN, POUT, FL>(wbe, Wa, Xa, split_a, Wb, Xb, sBias, sY, sRes, sMask, N, wave, lane, cx, pf, nextW, gPre,
                                                         sMaxOut);
  } else if constexpr (GN != 0) {
    node_gemm_n<HP, EPI, true, 3>(wb, Wa, XaS, Wb, XbS, sBias, sY, sRes, sMask, N, wave, lane, tw, &pf, nextW, gPre);
  } else {
    node_gemm<HP, EPI, true>(wb, Wa, Xa, Wb, Xb, sBias, sY, sRes, sMask, N, wave, lane, tw, &pf, nextW, gPre);
  }
}
// where the split copies of a node phase go: the weight ring's free slot (a resident kernel's other slot holds the first group of
// the next edge GEMM, requested by the previous one's last trip), or the whole ring where it idles across node phases (RI: the GN
// kernels and the half-ring mode, whose slot alone is too small).  cap: floats of that region; two inputs side by side if they fit.
// (shared with the host's LDS plan, gaudi_hip.hip: node_f16_fits)
__host__ __device__ constexpr bool node_ring_idle(int HP, int SP, bool GN) {
  return GN || (SP != 0 && GAUDI_NODE_F16 != 0 && (SP == 2 || HP < 48));
}
template <int HP>
__device__ __forceinline__ NodeCtxH node_ctx_h(float* region, int cap, int N, float winv, bool ktail, float* scales) {
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  const int need = nh_split_floats(HP, nct);
  return NodeCtxH{winv, region, 2 * need <= cap ? region + need : region, ktail, scales};
}
// Round 6: a KEPT split copy of h.  h feeds P and Q of a layer, the node MLP's first Linear of the same layer and -- in the denoiser,
// where the EquivariantUpdate leaves h alone (egnn_new.py:119-155) -- P and Q of the next sub-layer: with `keep` (an LDS region of
// its own behind everything else: nh_keep_floats, placed by the host when the plan leaves the room) h is split once per change
// instead of once per GEMM that reads it: 2 of the denoiser's 5 split passes per block, 1 of the predictor's 4 per layer.
// The context of a GEMM whose FIRST source is h: copy and factors in `keep`, a second source (agg) in the phase's usual region.
__host__ __device__ constexpr int nh_keep_floats(int HP, int N) { return nh_split_floats(HP, N <= 16 ? 1 : N <= 32 ? 2 : 3) + kScaleFloatsH; }
template <int HP>
__device__ __forceinline__ NodeCtxH node_ctx_keep(const NodeCtxH& cx, float* keep, int N) {
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  return NodeCtxH{cx.winv, keep + kScaleFloatsH, cx.split_a, cx.ktail, keep, cx.scales};
  (void)nct;
}

}  // namespace w8
}  // namespace gaudi
