// w8_edm.h -- EGNN_dynamics._forward (edm/egnn/models.py:83-152, edm/egnn/egnn_new.py) for one molecule held by one
// 8-wave workgroup.  Input z and output eps_hat live in LDS.  Weight buffer layout = EdmLayout (edm_device.h), matrices
// packed lane-linear (w8_common.h).
#pragma once
#include "edm_device.h"
#include "w8_common.h"
#include "w8_split.h"
#include "w8_nodes_f16.h"

namespace gaudi {
namespace w8 {

// LDS working set of one network evaluation.  GN ("global node buffers", the V8G kernels of round 4): the five [N][HP+4] node
// buffers -- what outgrows 160 KiB beyond ~22 graph nodes at the default widths -- live in a per-workgroup global scratch
// instead (the 4-wave family's V4G form of round 3, edm_device.h, on the 8-wave kernels).  The code is the same: the pointers
// are carved from the scratch, hipcc emits flat / global accesses for them; every cross-wave hand-off of these buffers
// already sits behind a workgroup barrier, which orders global memory inside a workgroup too.
// GN = 2 (round 6, hybrid residency: kern8gp_*.hip): of the five buffers the two that the EDGE phases gather from -- P and Q, read per
// edge slot and K chunk by the generating edge GEMM and again by the reverse pass's chain -- stay in LDS (up to N = 40 they fit
// beside the full ring: 2 x 34 KB + 52 KiB).  Measured on BASELINE config 4 read literally (N = 40, same session): 84.3 against
// 81.3 mol/s (+3.6 %) -- the gathers were hitting L1 / L2 well enough.  Complete graphs of 27-33 nodes (992 edge slots: 40 KB of
// per-slot arrays) do not fit that form: the host plans GN = 2 first and GN = 1 where it does not fit (gaudi_hip.hip: stage_graph8).
// node buffers a GN kernel keeps in LDS (shared with the host's LDS plan)
__host__ __device__ constexpr int gn_lds_buffers(int gn) { return gn == 2 ? 2 : 0; }
template <int HP, int SP = 0, int GN = 0>
struct NetSmem {
  static constexpr bool kGlobalNodes = GN != 0;
  float *h, *p, *q;     // [N][HP+4]
  float *agg, *agg1;    // [N][HP+4] the two partial edge->node sums of a node (its run may straddle two tiles)
  float* ring;          // weight ring of the edge GEMMs (EdgeRing<HP, SP>::kFloats)
  float *x, *x0;        // [N][4]
  uint32_t *pmax, *qmax;  // [N] bits of max |P_n|, max |Q_n| over the features (the split edge GEMMs' column scales, w8_split.h)
  float* hsc;             // [2][kScaleFloatsH] per-node descale factors of the fp16 node GEMMs' inputs (w8_nodes_f16.h)
  f4* geo;              // [S] (r, dhat)
  float* d0;            // [S]
  float* trans;         // [S][4]
  float* vec;           // [8*HP] the current layer's small vectors (cr, cd, b1, b2, wa/w3, bn1, bn2, ba)
  float* hk = nullptr;  // kept split copy of h (w8_nodes_f16.h: node_ctx_keep), behind the whole plan; nullptr: none
  __host__ __device__ static int floats(int N, int S) {
    return (GN ? gn_lds_buffers(GN) : 5) * N * (HP + 4) + EdgeRing<HP, SP>::kFloats + 8 * N + 2 * align4(N) + 96 + S * 9 + 8 * HP;
  }
  __device__ void carve(float* base, int N, int S, float* gnode = nullptr) {
    constexpr int LD = HP + 4;
    ring = base; base += EdgeRing<HP, SP>::kFloats;   // first: 1 KiB tiles stay 16-byte aligned whatever N is
    if (GN) gnode = assume_global(gnode);
    float*& nb = GN ? gnode : base;
    h = nb; nb += N * LD;
    if (GN == 2) {
      p = base; base += N * LD;
      q = base; base += N * LD;
    } else {
      p = nb; nb += N * LD;
      q = nb; nb += N * LD;
    }
    agg = nb; nb += N * LD;
    agg1 = nb; nb += N * LD;
    x = base; base += 4 * N;
    x0 = base; base += 4 * N;
    pmax = (uint32_t*)base; base += align4(N);  // (whole float4s: everything behind stays 16-byte aligned -- geo, trans and the
    qmax = (uint32_t*)base; base += align4(N);  // layer vectors are read with ds_read_b128)
    hsc = base; base += 96;
    geo = (f4*)base; base += S * 4;
    d0 = base; base += S;
    trans = base; base += S * 4;
    vec = base;
  }
};

// r = |x_i - x_j|^2, dhat = (x_i - x_j) / (sqrt(r + 1e-8) + norm_constant)   (egnn_new.py:394-400)
template <class SM>
__device__ __forceinline__ void compute_geo(const SM& sm, const MolGraph& mg, float norm_constant, int tid, bool write_d0) {
  for (int slot = tid; slot < mg.ntiles * 16; slot += kThreads) {
    const uint32_t e = mg.edge[slot];
    const int i = ew_i(e), j = ew_j(e);
    const float dx = sm.x[4 * i + 0] - sm.x[4 * j + 0];
    const float dy = sm.x[4 * i + 1] - sm.x[4 * j + 1];
    const float dz = sm.x[4 * i + 2] - sm.x[4 * j + 2];
    const float r = dx * dx + dy * dy + dz * dz;
    if (write_d0) {
      sm.d0[slot] = r;
    } else {
      const float inv = 1.0f / (sqrtf(r + 1e-8f) + norm_constant);
      sm.geo[slot] = (f4){r, dx * inv, dy * inv, dz * inv};
    }
  }
}

// x <- (x + sum_j trans_ij / normf) * mask     (egnn_new.py:132-155), fixed ascending-j order
template <class SM>
__device__ __forceinline__ void coord_update(const SM& sm, const MolGraph& mg, float normf, int tid) {
  for (int idx = tid; idx < mg.N * 3; idx += kThreads) {  // strided: small hidden sizes fit N > 170 nodes in LDS
    const int n = idx / 3, d = idx % 3;
    const uint32_t sg = mg.seg[n];
    const int st = sg >> 16, len = sg & 0xffff;
    float s = 0.f;
    for (int k = 0; k < len; ++k) s += sm.trans[(st + k) * 4 + d];
    sm.x[4 * n + d] = (sm.x[4 * n + d] + s / normf) * mg.mask[n];
  }
}

// The lane's view of its 16-edge tile in one round
struct TileCols {
  bool active;      // the wave owns a tile in this round
  int slot;         // the lane's slot (tile * 16 + column)
  int i, j;         // receiving / sending node of the lane's column
  bool run_end;     // last column of its run inside the tile
  int part;         // which partial buffer the run's sum goes to
  float mk;         // edge_mask value
  RunMask rm;
};
__device__ __forceinline__ TileCols load_tile(const MolGraph& mg, int round, int wave, int c) {
  TileCols tc;
  const int tile = round * kWaves + wave;
  tc.active = __builtin_amdgcn_readfirstlane(tile < mg.ntiles ? 1 : 0) != 0;
  tc.slot = (tc.active ? tile : 0) * 16 + c;
  const uint32_t e = mg.edge[tc.slot];
  tc.i = ew_i(e);
  tc.j = ew_j(e);
  tc.run_end = ew_run_end(e);
  tc.part = ew_part(e);
  tc.mk = mg.em[tc.slot];
  tc.rm.set(c, ew_run_start(e));
  return tc;
}

// edge -> node sums of one tile: segmented scan down the 16 columns, the last column of each run stores the run's sum into
// the node's row of partial buffer `part` (a node's run touches at most two tiles: host invariant)
template <int HP>
__device__ __forceinline__ void scatter_runs(f4 (&e)[HP / 16], const TileCols& tc, float* agg0, float* agg1, int g) {
  constexpr int T = HP / 16, LD = HP + 4;
  float* dst = (tc.part ? agg1 : agg0) + tc.i * LD + 4 * g;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const f4 s = seg_scan(e[t], tc.rm);
    if (tc.run_end) *(f4*)(dst + 16 * t) = s;
  }
}

// eps_hat[N][D] (LDS) = EGNN_dynamics._forward(t, z[N][D] (LDS))
template <int HP, int SP = 0, int GN = 0, bool FL = false>
__device__ __forceinline__ void edm_forward(const EdmDev& W, const MolGraph& mg, const NetSmem<HP, SP, GN>& sm, const float* sZ,
                                            float* sEps, float* sMean /* [4] */, float t_val, int tid STAMP_DECL) {
  constexpr int LD = HP + 4;
  constexpr int T = HP / 16;
  constexpr int PK = HP * HP;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int N = mg.N, D = mg.D, F = W.F, F1 = F + 1;
  EdmLayout lay{HP, F1, W.L, W.S};
  const float* __restrict__ w = W.w;
  const WBuf wb = make_wbuf(W.w, W.w_bytes);
  const WBuf wbe = SP ? make_wbuf(W.ws, W.ws_bytes) : wb;  // edge-GEMM matrices (w8_split.h)
  const bool tw = W.ktail != 0;                            // H % 16 == 4: the node GEMMs' tail tile (w8_common.h: tail_lane)

  // ---- input split + masking (models.py:88-105): x = z[:, :3]*m ; h = [z[:, 3:]*m , t]
  for (int idx = tid; idx < N * 3; idx += kThreads) {
    const int n = idx / 3, d = idx % 3;
    const float v = sZ[n * D + d] * mg.mask[n];
    sm.x[4 * n + d] = v;
    sm.x0[4 * n + d] = v;
  }
  // ---- embedding Linear(F+1 -> H)  (egnn_new.py:304)
  {
    const float* ew = w + lay.emb_w();
    const float* eb = w + lay.emb_b();
    for (int idx = tid; idx < N * HP; idx += kThreads) {
      const int n = idx / HP, f = idx % HP;
      float acc = 0.f;
      const float m = mg.mask[n];
      for (int k = 0; k < F; ++k) acc += gload(ew + f * F1 + k) * (sZ[n * D + 3 + k] * m);
      acc += gload(ew + f * F1 + F) * t_val;
      sm.h[n * LD + f] = acc + gload(eb + f);
    }
  }
  __syncthreads();
  compute_geo(sm, mg, 0.f, tid, true);  // d0 of the input coordinates (egnn_new.py:301)
  typename EdgeRing<HP, SP>::type ring;
  er_init<HP>(ring, sm.ring, W.ktail != 0, W.ws, W.hinv);
  static_assert(SP == 0 || NodeMath<SP>::kF16, "the split edge GEMMs take their column scales from the fp16 node GEMMs' row maxima");
  // NH: node GEMMs on fp16 pairs (w8_nodes_f16.h); their inputs' split copies live in the weight ring during the node phases
  constexpr bool NH = NodeMath<SP>::kF16;
  // RI: the ring is idle during the node phases -- GN: the node GEMMs' input rows are staged in it (w8_common.h: stage_rows; the
  // fp16 form's split copies instead); half-ring mode (and the smallest hidden size) with the fp16 form: the free slot alone
  // would not hold a split copy -- and every edge phase requests its first weight group itself instead of having it travel
  // across the node phase
  constexpr bool RI = node_ring_idle(HP, SP, GN);
  constexpr bool STG = GN && !NH;  // fp32 node GEMMs of a GN kernel read staged rows
  float* const xs0 = sm.ring;
  float* const xs1 = sm.ring + stage_stride(N * LD);
  if constexpr (!RI) er_start<HP>(ring, wbe, lay.gcl(0, 0) + 2 * PK, wave, lane);  // first edge GEMM: W2 of block 0's first GCL
  typename NodePFSel<HP, NH>::type pf;  // first weight tiles of the next node GEMM, loaded ahead of it
  node_prefetch_x<HP, NH, kAheadOne>(pf, wb, wbe, lay.gcl(0, 0), mg.NC, wave, lane, tw);
  // split-copy region of the node GEMMs of this phase (NH)
  auto hctx = [&]() {
    if constexpr (NH) {
      if constexpr (RI) return node_ctx_h<HP>(sm.ring, EdgeRing<HP, SP>::kFloats, mg.NC, W.hinv, tw, sm.hsc);
      else return node_ctx_h<HP>(ring.slot(ring.par ^ 1), EdgeRing<HP, SP>::kFloats / 2, mg.NC, W.hinv, tw, sm.hsc);
    } else {
      return NodeCtxH{1.f, nullptr, nullptr, tw, nullptr};
    }
  };
  // a kept split copy of h (w8_nodes_f16.h: node_ctx_keep): h is split when it has changed, not by every GEMM that reads it
  const bool keep = NH && sm.hk != nullptr;
  bool h_kept = false;
  constexpr int NV = (7 * HP + 16 + kThreads - 1) / kThreads;
  VecPF<NV> vpf;  // the next sub-layer's vectors (GCL: 7 HP + 16 floats, EquivariantUpdate: 5 HP), loaded a phase ahead
  vec_prefetch<NV, kThreads>(vpf, wb, lay.gcl(0, 0) + 6 * PK, 7 * HP + 16, tid);
  STAMP(ST_EDM_IO);

  for (int l = 0; l < W.L; ++l) {
    compute_geo(sm, mg, W.norm_constant, tid, false);  // egnn_new.py:216
    STAMP(ST_GEO);
    for (int s = 0; s < W.S; ++s) {
      // ------------------------------------------------------------------ GCL (egnn_new.py:42-89)
      const int G = lay.gcl(l, s);  // float offsets into the weight buffer
      const int Wnext_edge = s + 1 < W.S ? lay.gcl(l, s + 1) + 2 * PK : lay.equ(l) + 2 * PK;
      if constexpr (STG) stage_rows(xs0, sm.h, N * LD, wave, lane);
      for (int idx = tid; idx < N * (LD / 4); idx += kThreads) {
        *(f4*)(sm.agg + 4 * idx) = splat(0.f);
        *(f4*)(sm.agg1 + 4 * idx) = splat(0.f);
      }
      if (tid < N) {
        sm.pmax[tid] = 0u;
        sm.qmax[tid] = 0u;
      }
      vec_commit<NV, kThreads>(vpf, sm.vec, 7 * HP + 16, tid);  // (last: the wait for the vectors overlaps the stores above)
      if constexpr (STG) stage_wait();
      else __syncthreads();
      STAMP(ST_STAGE);
      const float *cr = sm.vec, *cd = sm.vec + HP, *b1 = sm.vec + 2 * HP, *b2 = sm.vec + 3 * HP, *wa = sm.vec + 4 * HP,
                  *bn1 = sm.vec + 5 * HP, *bn2 = sm.vec + 6 * HP;
      const float ba = sm.vec[7 * HP];
      const float crmax = sm.vec[7 * HP + 1], cdmax = sm.vec[7 * HP + 2];  // max |c_r|, max |c_d| (host)
      {
        const NodeCtxH cx = keep ? node_ctx_keep<HP>(hctx(), sm.hk, mg.NC) : hctx();
        node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadOne, kAheadAll, FL>(wb, wbe, G, sm.h, xs0, !h_kept, -1, nullptr, nullptr, b1, sm.p, nullptr, nullptr, mg.NC,
                                                      wave, lane, tw, cx, pf, G + PK, nullptr, sm.pmax);
        h_kept = keep;
        node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadAll, kAheadOne, FL>(wb, wbe, G + PK, sm.h, xs0, false, -1, nullptr, nullptr, nullptr, sm.q, nullptr,
                                                      nullptr, mg.NC, wave, lane, tw, cx, pf,
                                                      G + 3 * PK, nullptr, sm.qmax);  // node MLP weights travel across the edge phase
      }
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
      if constexpr (RI) er_start<HP>(ring, wbe, G + 2 * PK, wave, lane);
      for (int rd = 0; rd < mg.rounds; ++rd) {
        const TileCols tc = load_tile(mg, rd, wave, c);
        const f4 gg = sm.geo[tc.slot];
        f4 acc[T];
        const float d0v = sm.d0[tc.slot];
        const float ub = __builtin_bit_cast(float, sm.pmax[tc.i]) + __builtin_bit_cast(float, sm.qmax[tc.j]) + crmax * gg[0] + cdmax * fabsf(d0v);
        er_gemm_pq<HP>(acc, ring, wbe, G + 2 * PK, rd + 1 < mg.rounds ? G + 2 * PK : (RI ? -1 : Wnext_edge), b2, cr, cd,
                         sm.p + tc.i * LD + 4 * g, sm.q + tc.j * LD + 4 * g, gg[0], d0v, ub, tc.active, wave, lane STAMP_ARGS);
        STAMP(ST_EDGE);
        if (tc.active) {
          float sdot = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f4 m = silu4v(acc[t]);
            acc[t] = m;
            const f4 wv = *(const f4*)(wa + 16 * t + 4 * g);
            sdot += m[0] * wv[0] + m[1] * wv[1] + m[2] * wv[2] + m[3] * wv[3];
          }
          float a = 1.f;
          if (W.attention) a = sigmoid_f(reduce_groups(sdot) + ba);
          const float sc = a * tc.mk;
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = acc[t] * sc;
          scatter_runs<HP>(acc, tc, sm.agg, sm.agg1, g);
        }
        STAMP(ST_EDGE_EPI);
      }
      __syncthreads();
      STAMP(ST_BARRIER);
      if constexpr (STG) stage_rows(xs0, sm.h, N * LD, wave, lane);
      for (int idx = tid; idx < N * (HP / 4); idx += kThreads) {  // agg = (partial 0 + partial 1) / normalization_factor
        const int n = idx / (HP / 4), f = 4 * (idx % (HP / 4));
        if constexpr (STG)  // straight into the staged copy (only this GEMM reads agg)
          *(f4*)(xs1 + n * LD + f) = (*(const f4*)(sm.agg + n * LD + f) + *(const f4*)(sm.agg1 + n * LD + f)) / W.normf;
        else
          *(f4*)(sm.agg + n * LD + f) = (*(const f4*)(sm.agg + n * LD + f) + *(const f4*)(sm.agg1 + n * LD + f)) / W.normf;
      }
      if constexpr (STG) stage_wait();
      else __syncthreads();
      STAMP(ST_MISC);
      node_gemm_x<HP, EPI_SILU, true, GN, NH, kAheadOne, kAheadOne, FL>(wb, wbe, G + 3 * PK, sm.h, xs0, !h_kept, G + 4 * PK, sm.agg, xs1, bn1, sm.p, nullptr, nullptr,
                                              mg.NC, wave, lane, tw, keep ? node_ctx_keep<HP>(hctx(), sm.hk, mg.NC) : hctx(), pf, G + 5 * PK);
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
      vec_prefetch<NV, kThreads>(vpf, wb, s + 1 < W.S ? lay.gcl(l, s + 1) + 6 * PK : lay.equ(l) + 3 * PK,
                                 s + 1 < W.S ? 7 * HP + 16 : 5 * HP + 16, tid);
      if constexpr (STG) {
        stage_rows(xs0, sm.p, N * LD, wave, lane);
        stage_wait();
      }
      node_gemm_x<HP, EPI_RESIDUAL_MASK, false, GN, NH, kAheadOne, kAheadOne, FL>(wb, wbe, G + 5 * PK, sm.p, xs0, true, -1, nullptr, nullptr, bn2, sm.h, sm.h, mg.mask,
                                                       mg.NC, wave, lane, tw, hctx(), pf,
                                                       s + 1 < W.S ? lay.gcl(l, s + 1) : lay.equ(l));
      h_kept = false;  // (h has changed)
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
    }
    // -------------------------------------------------------- EquivariantUpdate (egnn_new.py:119-155)
    {
      const int E = lay.equ(l);
      const int Wnext_edge = l + 1 < W.L ? lay.gcl(l + 1, 0) + 2 * PK : -1;
      vec_commit<NV, kThreads>(vpf, sm.vec, 5 * HP + 16, tid);
      if (tid < N) {
        sm.pmax[tid] = 0u;
        sm.qmax[tid] = 0u;
      }
      if constexpr (STG) {
        stage_rows(xs0, sm.h, N * LD, wave, lane);
        stage_wait();
      } else {
        __syncthreads();
      }
      STAMP(ST_STAGE);
      const float *cr = sm.vec, *cd = sm.vec + HP, *b1 = sm.vec + 2 * HP, *b2 = sm.vec + 3 * HP, *w3 = sm.vec + 4 * HP;
      const float crmax = sm.vec[5 * HP], cdmax = sm.vec[5 * HP + 1];  // max |c_r|, max |c_d| (host)
      {
        const NodeCtxH cx = keep ? node_ctx_keep<HP>(hctx(), sm.hk, mg.NC) : hctx();
        node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadOne, kAheadAll, FL>(wb, wbe, E, sm.h, xs0, !h_kept, -1, nullptr, nullptr, b1, sm.p, nullptr, nullptr, mg.NC,
                                                      wave, lane, tw, cx, pf, E + PK, nullptr, sm.pmax);
        h_kept = keep;  // (the EquivariantUpdate leaves h alone: the next block's GCL reads the same copy)
        node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadAll, kAheadOne, FL>(wb, wbe, E + PK, sm.h, xs0, false, -1, nullptr, nullptr, nullptr, sm.q, nullptr,
                                                      nullptr, mg.NC, wave, lane, tw, cx, pf, l + 1 < W.L ? lay.gcl(l + 1, 0) : -1, nullptr,
                                                      sm.qmax);
      }
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
      if constexpr (RI) er_start<HP>(ring, wbe, E + 2 * PK, wave, lane);
      for (int rd = 0; rd < mg.rounds; ++rd) {
        const TileCols tc = load_tile(mg, rd, wave, c);
        const f4 gg = sm.geo[tc.slot];
        f4 acc[T];
        const float d0v = sm.d0[tc.slot];
        const float ub = __builtin_bit_cast(float, sm.pmax[tc.i]) + __builtin_bit_cast(float, sm.qmax[tc.j]) + crmax * gg[0] + cdmax * fabsf(d0v);
        er_gemm_pq<HP>(acc, ring, wbe, E + 2 * PK, rd + 1 < mg.rounds ? E + 2 * PK : (RI ? -1 : Wnext_edge), b2, cr, cd,
                         sm.p + tc.i * LD + 4 * g, sm.q + tc.j * LD + 4 * g, gg[0], d0v, ub, tc.active, wave, lane STAMP_ARGS);
        STAMP(ST_EDGE);
        if (tc.active) {
          float sdot = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f4 m = silu4v(acc[t]);
            const f4 wv = *(const f4*)(w3 + 16 * t + 4 * g);
            sdot += m[0] * wv[0] + m[1] * wv[1] + m[2] * wv[2] + m[3] * wv[3];
          }
          const float phi = reduce_groups(sdot);
          const float tau = (W.use_tanh ? tanhf(phi) * W.coords_range : phi) * tc.mk;
          if (g == 0) *(f4*)(sm.trans + 4 * tc.slot) = (f4){gg[1] * tau, gg[2] * tau, gg[3] * tau, 0.f};
        }
        STAMP(ST_EDGE_EPI);
      }
      if (l + 1 < W.L) vec_prefetch<NV, kThreads>(vpf, wb, lay.gcl(l + 1, 0) + 6 * PK, 7 * HP + 16, tid);
      __syncthreads();
      STAMP(ST_BARRIER);
      coord_update(sm, mg, W.normf, tid);
      __syncthreads();
      STAMP(ST_MISC);
    }
  }

  // ---- head: embedding_out * mask (egnn_new.py:316-318); vel = (x - x_in) * mask, masked mean removal
  //      (models.py:116-152); the time column of h is dropped.
  {
    const float* ow = w + lay.out_w();
    const float* ob = w + lay.out_b();
    int bad = 0;
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      const float v = (sm.x[4 * n + d] - sm.x0[4 * n + d]) * mg.mask[n];
      bad += v != v;
      sEps[n * D + d] = v;
    }
    const float* hh = sm.h;
    const float* msk = mg.mask;
    small_dots<kThreads>(
        N * F, HP, tid, [=](int p, int k) { return gload(ow + (p % F) * HP + k); },
        [=](int p, int k) { return hh[(p / F) * LD + k]; },
        [=](int p, float acc) { sEps[(p / F) * D + 3 + p % F] = (acc + gload(ob + p % F)) * msk[p / F]; });
    // `if torch.any(torch.isnan(vel)): vel = torch.nan_to_num(vel, 0.0)` (models.py:138-141), triggered per molecule
    // (= per component of a packed graph)
    int badmask = 0;
    if (bad)
      for (int idx = tid; idx < N * 3; idx += kThreads) {
        const float v = sEps[(idx / 3) * D + idx % 3];
        if (v != v) badmask |= 1 << mg_comp(mg, idx / 3);
      }
    badmask = block_or_bits(badmask & 15, mg.ncomp);
    if (badmask) {
      for (int idx = tid; idx < N * 3; idx += kThreads) {
        const int n = idx / 3, d = idx % 3;
        const float v = sEps[n * D + d];
        if ((badmask >> mg_comp(mg, n)) & 1)
          sEps[n * D + d] = v != v ? 0.f : fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
      }
      __syncthreads();
    }
    if (tid < 3 * mg.ncomp) {
      const int k = tid / 3, d = tid % 3;
      float s = 0.f, cnt = 0.f;
      for (int n = 0; n < N; ++n)
        if (mg_comp(mg, n) == k) {
          s += sEps[n * D + d];
          cnt += mg.mask[n];
        }
      sMean[4 * k + d] = s / fmaxf(cnt, 1.0f);
    }
    __syncthreads();
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      const int k = mg_comp(mg, n);
      sEps[n * D + d] = sEps[n * D + d] - (k < mg.ncomp ? sMean[4 * k + d] : 0.f) * mg.mask[n];
    }
    __syncthreads();
  }
  STAMP(ST_EDM_IO);
}

}  // namespace w8
}  // namespace gaudi
