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
// scrubs both the same way (models.py:138-141).  The host refuses the form (-> fp32 node GEMMs) for weight sets with
// infinities or a matrix that lies more than 2^24 below the largest one.
//
// Bytes.  hi and lo are 2 + 2 bytes: a matrix image is exactly the fp32 matrix's size (units of 1 KiB ordered
// [K chunk of 32][output tile][piece], lane L = (row L & 15, inputs 8 (L >> 4) .. +7 of the chunk); a K tail -- H % 16 == 4 --
// is a trailing block of T x 256 B fp32 k-steps, as in the split edge images).  Images live at float offset 2 W of their own
// buffer (W = the fp32 buffer's offset; an odd tile count without a K tail needs (T + 1) / T of the fp32 size).
//
// Schedule.  Output tile t -> wave t & 7 (as the fp32 form).  The activations of a GEMM input are split ONCE by all waves into
// LDS (wave w: rows w, w + 8, ...; one DPP max-scan per row, no cross-wave step) in B-operand order -- two conflict-free
// ds_read_b128 per chunk and column tile; the weight stream runs kDepthH chunks ahead in registers, across the two sources of
// a GEMM and across calls (the next matrix's first chunks travel while this one drains).  Accumulation order per output element:
// K chunks in order, per chunk w_hi x_lo, w_hi x_hi, w_lo x_hi; then the K tail's fp32 step; sources in order -- independent of
// the column-tile count and of the wave's tile count (packed launches stay bit-identical to unpacked ones).
#pragma once
#include "w8_split.h"

namespace gaudi {
namespace w8 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) uint32_t u2;

constexpr int kDepthH = 3;        // weight chunks in flight per wave
constexpr float kLoScale = 2048.f;  // 2^11

__device__ __forceinline__ f4 mfma_h(const u4 a, const u4 b, const f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u4 ldu4h(const WBuf& wh, int off_floats, int lane) {
  return __builtin_bit_cast(u4, ldw4n(wh, off_floats, lane));
}

// ---- geometry shared with the host packer (gaudi_hip.hip: pack_matrix_f16)
__host__ __device__ constexpr bool nh_has_tail(int HP, bool ktail) { return ktail && ((HP / 16) & 1) && HP / 16 >= 3; }
__host__ __device__ constexpr int nh_chunks(int HP, bool ktail) { return nh_has_tail(HP, ktail) ? (HP / 16 - 1) / 2 : (HP / 16 + 1) / 2; }
__host__ __device__ constexpr int nh_chunk_floats(int HP) { return (HP / 16) * 2 * 256; }             // one K chunk of an image
__host__ __device__ constexpr int nh_image_floats(int HP, bool ktail) {                               // <= 2 HP^2
  return nh_chunks(HP, ktail) * nh_chunk_floats(HP) + (nh_has_tail(HP, ktail) ? (HP / 16) * 64 : 0);
}
// split copy of one GEMM input in LDS, NCT column tiles of 16 nodes: [chunk][column tile][piece][64 x 16 B] with 64 B of padding
// per chunk (the splitting wave's 8-byte stores of one row then fall on distinct banks), then the K tail's fp32 B operands
// [column tile][64] and the per-node descale factors [column tile][16]
__host__ __device__ constexpr int nh_chunk_stride(int nct) { return nct * 512 + 16; }
__host__ __device__ constexpr int nh_split_floats(int HP, int nct) { return ((HP / 16 + 1) / 2) * nh_chunk_stride(nct) + nct * 80; }

struct SplitBufH {
  float* base;
  int nct;  // column tiles
  __device__ __forceinline__ float* chunk(int m) const { return base + m * nh_chunk_stride(nct); }
  __device__ __forceinline__ float* tail(int HP) const { return base + ((HP / 16 + 1) / 2) * nh_chunk_stride(nct); }
  __device__ __forceinline__ float* scale(int HP) const { return tail(HP) + nct * 64; }
};

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
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

__device__ __forceinline__ uint32_t pk_f16(float a, float b) {  // round to nearest even
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a, b}, h2));
}
__device__ __forceinline__ f2 unpk_f16(uint32_t p) { return __builtin_convertvector(__builtin_bit_cast(h2, p), f2); }

// all waves: rows [0, N) of X ([N][HP+4], LDS -- or global for the kernels with node buffers in global memory) -> split
// B operands.  Wave w owns rows w, w + 8, ...: lane j holds inputs 4 j .. 4 j + 3 of the row.  The caller places a barrier
// between this and the GEMM's reads.  Columns >= N of the last tile are left as they are: a matrix-instruction column depends
// on its own B column only and those results are never stored.
template <int HP>
__device__ __forceinline__ void split_rows_h(const SplitBufH& sb, const float* X, int N, bool ktail, int wave, int lane) {
  constexpr int T = HP / 16, LD = HP + 4;
  const bool tail = nh_has_tail(HP, ktail);
  const int nc = nh_chunks(HP, ktail);
  const int m = lane >> 3, g = (lane >> 1) & 3, half = lane & 1;
  float* const tl = sb.tail(HP);
  float* const sc = sb.scale(HP);
  for (int n = wave; n < N; n += kWaves) {
    const f4 x = lane < HP / 4 ? *(const f4*)(X + n * LD + 4 * lane) : splat(0.f);
    const uint32_t a0 = __builtin_bit_cast(uint32_t, x[0]) & 0x7fffffffu, a1 = __builtin_bit_cast(uint32_t, x[1]) & 0x7fffffffu,
                   a2 = __builtin_bit_cast(uint32_t, x[2]) & 0x7fffffffu, a3 = __builtin_bit_cast(uint32_t, x[3]) & 0x7fffffffu;
    const uint32_t mx = wave_max_u32(umax(umax(a0, a1), umax(a2, a3)));
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
      float* d = sb.chunk(m) + ct * 512 + (4 * c + g) * 4 + half * 2;
      *(u2*)d = (u2){h01, h23};
      *(u2*)(d + 256) = (u2){l01, l23};
    }
    if (tail && lane == 4 * (T - 1)) *(f4*)(tl + ct * 64 + 4 * c) = x;  // inputs 16 (T-1) .. +3, unscaled: the fp32 k-step
    if (lane == 0) sc[ct * 16 + c] = inv;
  }
}

// LDS writes of every wave visible to every wave; global loads in flight stay in flight (__syncthreads would wait for them)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// one chunk of the wave's (up to two) output tiles: [tile][piece]
struct NodeSetH {
  u4 p[2][2];
};
template <int HP>
struct NodePFH {
  NodeSetH s[kDepthH];  // chunks 0 .. kDepthH-1 of the next node GEMM, loaded ahead of the call
};

template <int NTW>
__device__ __forceinline__ void nh_load(NodeSetH& s, const WBuf& wh, int chunk_off, int wave, int lane) {
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    s.p[u][0] = ldu4h(wh, chunk_off + ((wave + kWaves * u) * 2 + 0) * 256, lane);
    s.p[u][1] = ldu4h(wh, chunk_off + ((wave + kWaves * u) * 2 + 1) * 256, lane);
  }
}
template <int HP>
__device__ __forceinline__ void node_prefetch_h(NodePFH<HP>& pf, const WBuf& wh, int W /* fp32 float offset */, int wave, int lane) {
  constexpr int T = HP / 16;
  if (wave + kWaves < T) {
#pragma unroll
    for (int d = 0; d < kDepthH; ++d) nh_load<2>(pf.s[d], wh, 2 * W + d * nh_chunk_floats(HP), wave, lane);
  } else if (wave < T) {
#pragma unroll
    for (int d = 0; d < kDepthH; ++d) nh_load<1>(pf.s[d], wh, 2 * W + d * nh_chunk_floats(HP), wave, lane);
  }
}

// what a call needs beside the fp32 form's arguments
struct NodeCtxH {
  float winv;       // 2^-s_w: descale of the network's node matrices
  float* split_a;   // LDS: split copy of the first source (nh_split_floats(HP, nct) floats)
  float* split_b;   // ... of the second source; == split_a: the region holds ONE input, the sources are split in turn
  bool ktail;
};

template <int HP, int EPI, int NT, int NTW>
__device__ __forceinline__ void node_gemm_h_body(const WBuf& wh, int Wa, const float* sXa, bool do_split_a, int Wb, const float* sXb,
                                                 const float* sBias, float* sY, const float* sRes, const float* sMask, int N, int wave,
                                                 int lane, const NodeCtxH& cx, NodePFH<HP>& pf, int nextW, float* gPre) {
  constexpr int T = HP / 16, LD = HP + 4;
  const int c = lane & 15, g = lane >> 4;
  const bool tail = nh_has_tail(HP, cx.ktail);
  const int nc = nh_chunks(HP, cx.ktail);
  const int KT = Wb >= 0 ? 2 * nc : nc;  // two sources run as ONE K loop so the load pipeline never restarts
  const bool seq = Wb >= 0 && cx.split_b == cx.split_a;
  const SplitBufH sa{cx.split_a, NT}, sb{cx.split_b, NT};
  const int bpos = (4 * c + g) * 4;  // the lane's float offset inside a 1 KiB B unit
  // float offset (in the image buffer) of chunk cc of the stream Wa | Wb | nextW; past the end: clamped (surplus loads are unused)
  auto chunk_off = [&](int cc) {
    if (cc >= KT) {
      if (nextW >= 0) return 2 * nextW + (cc - KT < nc ? cc - KT : nc - 1) * nh_chunk_floats(HP);
      cc = KT - 1;
    }
    return cc < nc ? 2 * Wa + cc * nh_chunk_floats(HP) : 2 * Wb + (cc - nc) * nh_chunk_floats(HP);
  };
  struct BH {
    u4 h[NT], l[NT];
  };
  auto bld = [&](int cc) {
    const float* q = (cc < nc ? sa.chunk(cc) : sb.chunk(cc - nc)) + bpos;
    BH b;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      b.h[j] = *(const u4*)(q + j * 512);
      b.l[j] = *(const u4*)(q + j * 512 + 256);
    }
    return b;
  };
  // K-tail weights (one fp32 k-step per output tile and source) travel with the first chunks
  float ta[NTW], tb[NTW];
  if (tail) {
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int t = wave + kWaves * u;
      ta[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wh.r, lane * 4, (2 * Wa + nc * nh_chunk_floats(HP) + t * 64) * 4, 0));
      tb[u] = Wb >= 0 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wh.r, lane * 4, (2 * Wb + nc * nh_chunk_floats(HP) + t * 64) * 4, 0))
                      : 0.f;
    }
  }
  if (do_split_a) split_rows_h<HP>(sa, sXa, N, cx.ktail, wave, lane);
  if (Wb >= 0 && !seq) split_rows_h<HP>(sb, sXb, N, cx.ktail, wave, lane);
  if (do_split_a || (Wb >= 0 && !seq)) lds_barrier();

  f4 acc0[NT][NTW], acc1[NT][NTW], y[NT][NTW];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      acc0[j][u] = splat(0.f);
      acc1[j][u] = splat(0.f);
      y[j][u] = sBias != nullptr ? *(const f4*)(sBias + 16 * (wave + kWaves * u) + 4 * g) : splat(0.f);
    }
  // descale and fold the accumulators of one source into y
  auto fold = [&](const SplitBufH& s_, const float* X, const float (&tw)[NTW]) {
    if (tail) {  // the K tail's fp32 step: inputs 16 (T-1) + g on lane group g, unscaled operands
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float xb = s_.tail(HP)[j * 64 + 4 * c + g];
#pragma unroll
        for (int u = 0; u < NTW; ++u) y[j][u] = mfma1(tw[u], xb, y[j][u]);
      }
    }
    (void)X;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float sc = s_.scale(HP)[j * 16 + c] * cx.winv;
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        y[j][u] = y[j][u] + (acc0[j][u] + acc1[j][u] * (1.0f / kLoScale)) * sc;
        acc0[j][u] = splat(0.f);
        acc1[j][u] = splat(0.f);
      }
    }
  };
  auto mm = [&](const NodeSetH& s, const BH& b) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < NTW; ++u) acc1[j][u] = mfma_h(s.p[u][0], b.l[j], acc1[j][u]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < NTW; ++u) acc0[j][u] = mfma_h(s.p[u][0], b.h[j], acc0[j][u]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int u = 0; u < NTW; ++u) acc1[j][u] = mfma_h(s.p[u][1], b.h[j], acc1[j][u]);
  };
  BH bcur = bld(0);
  auto step = [&](NodeSetH& s, int cc) {  // consume chunk cc from s, refill s with chunk cc + kDepthH of the stream
    if (Wb >= 0 && cc == nc) {            // the second source begins
      fold(sa, sXa, ta);
      if (seq) {
        __syncthreads();  // (every wave is done with the first source's copy)
        split_rows_h<HP>(sb, sXb, N, cx.ktail, wave, lane);
        lds_barrier();
        bcur = bld(cc);
      }
    }
    const bool more = cc + 1 < KT && !(seq && cc + 1 == nc);
    BH bnext = bcur;
    if (more) bnext = bld(cc + 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(s, bcur);
    __builtin_amdgcn_sched_barrier(0);
    if (cc + kDepthH < KT || nextW >= 0) nh_load<NTW>(s, wh, chunk_off(cc + kDepthH), wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (more) bcur = bnext;
  };
  static_assert(kDepthH == 3, "the K loop below rotates three operand sets");
#pragma unroll 1
  for (int cc = 0; cc < KT; cc += 3) {
    step(pf.s[0], cc);
    if (cc + 1 < KT) step(pf.s[1], cc + 1);
    if (cc + 2 < KT) step(pf.s[2], cc + 2);
  }
  if (nextW >= 0) {  // the next matrix's chunk i sits in set (KT + i) % 3: bring chunk 0 to set 0
    const int r = KT % 3;
    if (r == 1) {
      const NodeSetH t0 = pf.s[0];
      pf.s[0] = pf.s[1]; pf.s[1] = pf.s[2]; pf.s[2] = t0;
    } else if (r == 2) {
      const NodeSetH t0 = pf.s[0];
      pf.s[0] = pf.s[2]; pf.s[2] = pf.s[1]; pf.s[1] = t0;
    }
  }
  if (Wb >= 0) fold(sb, sXb, tb);
  else fold(sa, sXa, ta);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int t = wave + kWaves * u;
      const int nd = j * 16 + c;
      f4 yy = y[j][u];
      if (nd < N) {
        float* dst = sY + nd * LD + 16 * t + 4 * g;
        if (tail && t == T - 1 && g > 0) yy = splat(0.f);  // rows 4 .. 15 of the tail tile are padding
        if (gPre != nullptr) stash_store((f4*)(gPre + nd * HP + 16 * t + 4 * g), yy);  // stash: write once, read once
        if (!(tail && t == T - 1 && g > 0)) {
          if (EPI == EPI_SILU) yy = silu4(yy);
          if (EPI == EPI_RESIDUAL_MASK) {
            const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g);
            yy = (r + yy) * sMask[nd];
          }
          if (EPI == EPI_MUL_DSILU) {  // y * silu'(pre-activation stored in sRes); in place is safe
            const f4 r = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g);
            yy = (f4){yy[0] * dsilu_f(r[0]), yy[1] * dsilu_f(r[1]), yy[2] * dsilu_f(r[2]), yy[3] * dsilu_f(r[3])};
          }
          if (EPI == EPI_ACCUM) yy = *(const f4*)(sRes + nd * LD + 16 * t + 4 * g) + yy;
        }
        *(f4*)dst = yy;
      }
    }
}

// One node GEMM of the workgroup (all waves call it; N <= 16 MAXNT).  do_split_a = false: an earlier call of this phase left
// the first source's split copy in cx.split_a (P and Q share h, the two transposed GEMMs of dnpre share it).  The caller
// guarantees that nobody still reads the split regions when the call starts (a barrier since their last use) and places a
// barrier between this call's stores to sY and their readers, as for the fp32 form.
template <int HP, int EPI, int MAXNT>
__device__ __forceinline__ void node_gemm_h(const WBuf& wh, int Wa, const float* sXa, bool do_split_a, int Wb, const float* sXb,
                                            const float* sBias, float* sY, const float* sRes, const float* sMask, int N, int wave,
                                            int lane, const NodeCtxH& cx, NodePFH<HP>& pf, int nextW = -1, float* gPre = nullptr) {
  constexpr int T = HP / 16;
  const bool two = wave + kWaves < T, one = wave < T;
  if (!one) {
    // a wave without an output tile (hidden sizes below 128): its share of the split and the barriers of the others
    const bool seq = Wb >= 0 && cx.split_b == cx.split_a;
    const int nct = N <= 16 ? 1 : (MAXNT < 3 || N <= 32) ? 2 : 3;
    if (do_split_a) split_rows_h<HP>(SplitBufH{cx.split_a, nct}, sXa, N, cx.ktail, wave, lane);
    if (Wb >= 0 && !seq) split_rows_h<HP>(SplitBufH{cx.split_b, nct}, sXb, N, cx.ktail, wave, lane);
    if (do_split_a || (Wb >= 0 && !seq)) lds_barrier();
    if (seq) {
      __syncthreads();
      split_rows_h<HP>(SplitBufH{cx.split_b, nct}, sXb, N, cx.ktail, wave, lane);
      lds_barrier();
    }
    return;
  }
  auto run = [&](auto nt_tag) {
    constexpr int NT = decltype(nt_tag)::value;
    if (two) node_gemm_h_body<HP, EPI, NT, 2>(wh, Wa, sXa, do_split_a, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, cx, pf, nextW, gPre);
    else node_gemm_h_body<HP, EPI, NT, 1>(wh, Wa, sXa, do_split_a, Wb, sXb, sBias, sY, sRes, sMask, N, wave, lane, cx, pf, nextW, gPre);
  };
  if (N <= 16) run(std::integral_constant<int, 1>{});
  else if (MAXNT < 3 || N <= 32) run(std::integral_constant<int, 2>{});
  else if constexpr (MAXNT >= 3) run(std::integral_constant<int, 3>{});
}

}  // namespace w8
}  // namespace gaudi
