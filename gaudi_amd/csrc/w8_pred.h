// w8_pred.h -- EGNN_predictor (edm/egnn_predictor/models.py:433-457,543-560; gcl.py:225-316) on the 8-wave kernels: forward
// with the per-layer activation stash, the hand-written reverse pass that replaces torch.autograd.grad at
// en_diffusion.py:900-903, and the guidance epilogue (:905-920).  One workgroup (8 waves) = one molecule; its live edges run in
// rounds of eight 16-slot tiles (one round for every molecule of the reference's datasets).  Weight layout = PredLayout
// (pred_device.h), matrices packed lane-linear.
#pragma once
#include "pred_device.h"
#include "w8_edm.h"

namespace gaudi {
namespace w8 {

// GN: the five node buffers live in a per-workgroup global scratch (w8_edm.h: NetSmem); the publish buffer of the reverse
// pass is then [ring | pub] -- split forms only
// PG (round 6, wide groups): ONE of the five node buffers -- b4: the second partial edge->node sums of the forward pass, dnpre / dQ in
// the reverse pass -- lives in the workgroup's global scratch, everything else as in the resident kernels (GN = 0).  Two cata-11
// molecules per workgroup (22 node slots, 256 edge slots) then fit beside the FULL weight ring at the default widths, where five
// resident buffers leave room for the half ring only (two trips per K chunk: -6.6 % measured on pairs that fit both,
// tools/ring_mode_ab.sh).
template <int HP, int SP = 0, int GN = 0, bool PG = false>
struct PredSmem {
  static_assert(!(PG && GN != 0), "PG is a variant of the resident kernels");
  static_assert(!GN || SP != 0, "global node buffers: split edge GEMMs only (the fp32 form's publish buffer starts in b0 / b1)");
  static constexpr bool kGlobalNodes = GN != 0;
  float *b2, *b3, *b4;            // [N][HP+4] node buffers (roles change per phase, see below)
  float *b0, *b1;                 // [N][HP+4] ... these two open the publish buffer of the reverse pass: [b0 | b1 | (ring) | pub]
  float* ring;                    // weight ring of the edge GEMMs (EdgeRing<HP, SP>::kFloats); split form: idle while du is
                                  // published (no group is prefetched across that phase), so it is part of the buffer too
  float* pub;                     // [pubx] extra floats of the publish buffer (reverse pass: du of every slot, CH tiles at a time)
  float *x, *x0, *dx;             // [N][4]
  uint32_t *pmax, *qmax;          // [N] bits of max |P_n|, max |Q_n| (w8_edm.h: NetSmem)
  float* hsc;                     // [2][kScaleFloatsH] (w8_edm.h: NetSmem)
  f4* geo;                        // [S]
  float *d0, *trans, *dd0;        // [S], [S][4], [S]
  float* pred;                    // [16] pred | [16] dpred
  float* vec;                     // [10*HP] the current layer's vectors (cr,cd,b1,b2,wa,bc1,wc2,bn1,bn2,ba)
  float* hk = nullptr;            // kept split copy of h (w8_nodes_f16.h: node_ctx_keep), behind the whole plan; nullptr: none
  __host__ __device__ static int floats(int N, int S, int pubx) {
    return EdgeRing<HP, SP>::kFloats + (GN ? gn_lds_buffers(GN) : PG ? 4 : 5) * N * (HP + 4) + pubx + 12 * N + 2 * align4(N) + 96 + S * 10 + 32 + 10 * HP;
  }
  // the publish buffer of the reverse pass (du of every slot, pub_ch feature tiles at a time)
  __device__ __forceinline__ float* publish() const { return GN ? ring : b0; }
  __device__ void carve(float* base, int N, int S, int pubx, float* gnode = nullptr) {
    constexpr int LD = HP + 4;
    if (SP == 0) { ring = base; base += EdgeRing<HP, SP>::kFloats; }  // fp32 form: the ring stays busy across the publish phase
    if (GN || PG) gnode = assume_global(gnode);
    float*& nb = GN ? gnode : base;
    // (GN = 2, round 6: P = b1 and Q = b2 -- what the edge phases of both passes gather from -- stay in LDS, w8_edm.h: gn_lds_buffers)
    if (GN == 2) { b2 = base; base += N * LD; } else { b2 = nb; nb += N * LD; }
    b3 = nb; nb += N * LD;
    if (PG) { b4 = gnode; gnode += N * LD; } else { b4 = nb; nb += N * LD; }
    b0 = nb; nb += N * LD;
    if (GN == 2) { b1 = base; base += N * LD; } else { b1 = nb; nb += N * LD; }
    if (SP != 0) { ring = base; base += EdgeRing<HP, SP>::kFloats; }  // N * LD * 4 bytes is a multiple of 16: units stay aligned
    pub = base; base += pubx;
    x = base; base += 4 * N;
    x0 = base; base += 4 * N;
    dx = base; base += 4 * N;
    pmax = (uint32_t*)base; base += align4(N);  // (whole float4s: w8_edm.h)
    qmax = (uint32_t*)base; base += align4(N);
    hsc = base; base += 96;
    geo = (f4*)base; base += S * 4;
    d0 = base; base += S;
    trans = base; base += S * 4;
    dd0 = base; base += S;
    pred = base; base += 32;
    vec = base;
  }
};

// stash per molecule: node part  L x { P [N][HP] | Q [N][HP] | npre [N][HP] | x [N][4] }
//                     edge part  L x (S/16 tiles) x { v | silu'(cpre) } x [HP/16][64 lanes] float4   (accumulator layout)
//                     gate part  L x S floats (attention gate a_ij of every slot) | L x S floats (phi_ij = wc2 . silu(cpre))
// The coordinate branch stashes what the reverse pass consumes -- silu'(cpre) and the scalar phi, both by-products of the
// forward's own sigmoid -- instead of cpre: the reverse pass then needs no transcendental on those 208 features per edge.
//                     du part    (S/16 tiles) x [HP/16][64 lanes] float4, graphs of more than one round (S > 128) only: the reverse
//                                pass parks every tile's du there between its chain and the publish phase
__host__ __device__ inline long long pred_stash_floats8(int N, int HP, int L, int S) {
  return pred_stash_node_floats(N, HP, L) + (long long)L * S * HP * 2 + 2LL * L * S + (S > 16 * kWaves ? (long long)S * HP : 0);
}
__device__ __forceinline__ size_t edge_stash_off8(int l, int tile, int arr, int S, int HP) {
  return (((size_t)l * (S / 16) + tile) * 2 + arr) * (size_t)(16 * HP);
}

// largest number of 16-feature tiles per publish chunk that fits `avail` floats for S slots (row = 16 CH + 4 floats)
__host__ __device__ inline int pub_chunk_tiles(int S, long long avail_floats, int T) {
  int ch = (int)((avail_floats / (S > 0 ? S : 1) - 4) / 16);
  return ch > T ? T : ch;
}

// ---------------------------------------------------------------------------------------------
// forward: pred[K] -> sm.pred[0..K)
// buffers: h = b0, P = b1, Q = b2, agg = b3, second agg partial = b4
// ---------------------------------------------------------------------------------------------
// MR: the kernel takes graphs of more than one round of eight edge tiles (more than 128 slots).  A separate instantiation: the
// round loops (and the second copy of the reverse chain that parks du in the stash) cost the single-round kernels 2-4 % when
// they live in the same function (hipcc's register allocation of the out-of-line phases changes), measured on C3.
template <int HP, int SP = 0, bool MR = false, int GN = 0, bool FL = false, class SM = PredSmem<HP, SP, GN>>
__device__ __forceinline__ void pred_forward(const PredDev& W, const MolGraph& mg, const SM& sm, const float* sZ,
                                             float t_val, float* stash, float readout_div, int tid STAMP_DECL) {
  constexpr int LD = HP + 4;
  constexpr int T = HP / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int N = mg.N, D = mg.D, F = W.F, F1 = F + 1, K = W.K, S = mg.S;
  PredLayout lay{HP, F1, K, W.L};
  const float* __restrict__ w = W.w;
  const WBuf wb = make_wbuf(W.w, W.w_bytes);
  const WBuf wbe = SP ? make_wbuf(W.ws, W.ws_bytes) : wb;  // edge-GEMM matrices (w8_split.h)
  const bool tw = W.ktail != 0;                            // H % 16 == 4: the node GEMMs' tail tile (w8_common.h: tail_lane)
  float *h = sm.b0, *p = sm.b1, *q = sm.b2, *agg = sm.b3, *agg1 = sm.b4;
  float* estash = stash + pred_stash_node_floats(N, HP, W.L);
  float* astash = estash + (size_t)W.L * S * HP * 2;
  float* pstash = astash + (size_t)W.L * S;

  for (int idx = tid; idx < N * 3; idx += kThreads) {  // models.py:439
    const int n = idx / 3, d = idx % 3;
    const float v = sZ[n * D + d] * mg.mask[n];
    sm.x[4 * n + d] = v;
    sm.x0[4 * n + d] = v;
  }
  {
    const float* ew = w + lay.emb_w();
    const float* eb = w + lay.emb_b();
    for (int idx = tid; idx < N * HP; idx += kThreads) {
      const int n = idx / HP, f = idx % HP;
      float acc = 0.f;
      const float m = mg.mask[n];
      for (int k = 0; k < F; ++k) acc += gload(ew + f * F1 + k) * (sZ[n * D + 3 + k] * m);
      acc += gload(ew + f * F1 + F) * t_val;
      h[n * LD + f] = acc + gload(eb + f);
    }
  }
  __syncthreads();
  compute_geo(sm, mg, 0.f, tid, true);  // edge_attr = |x_i - x_j|^2 of the input (models.py:452)
  typename EdgeRing<HP, SP>::type ring;
  er_init<HP>(ring, sm.ring, W.ktail != 0, W.ws, W.hinv);
  // NH / RI / STG: see edm_forward (w8_edm.h) -- fp16-pair node GEMMs; the ring idles across node phases (GN: node-GEMM inputs
  // are staged in it, every edge phase requests its first weight group itself); fp32 node GEMMs of a GN kernel read staged rows
  constexpr bool NH = NodeMath<SP>::kF16;
  constexpr bool RI = node_ring_idle(HP, SP, GN);
  constexpr bool STG = GN && !NH;
  float* const xs0 = sm.ring;
  float* const xs1 = sm.ring + stage_stride(N * LD);
  if constexpr (!RI) er_start<HP>(ring, wbe, lay.layer(0) + 2 * HP * HP, wave, lane);  // W2 of layer 0
  typename NodePFSel<HP, NH>::type pf;
  node_prefetch_x<HP, NH, kAheadOne>(pf, wb, wbe, lay.layer(0), mg.NC, wave, lane, tw);
  auto hctx = [&]() {
    if constexpr (NH) {
      if constexpr (RI) return node_ctx_h<HP>(sm.ring, EdgeRing<HP, SP>::kFloats, mg.NC, W.hinv, tw, sm.hsc);
      else return node_ctx_h<HP>(ring.slot(ring.par ^ 1), EdgeRing<HP, SP>::kFloats / 2, mg.NC, W.hinv, tw, sm.hsc);
    } else {
      return NodeCtxH{1.f, nullptr, nullptr, tw, nullptr};
    }
  };
  // a kept split copy of h: P splits it, Q and the node MLP's first Linear read the same copy (w8_nodes_f16.h: node_ctx_keep)
  const bool keep = NH && sm.hk != nullptr;
  constexpr int NV = (PredLayerW::vec_count(HP) + kThreads - 1) / kThreads;
  VecPF<NV> vpf;  // the next layer's vectors, loaded one node GEMM ahead
  vec_prefetch<NV, kThreads>(vpf, wb, PredLayerW::vec_off(lay.layer(0), HP), PredLayerW::vec_count(HP), tid);

  for (int l = 0; l < W.L; ++l) {
    const bool last = l == W.L - 1;  // the last layer's coordinate update never reaches the readout
    const PredLayerW Lw(w, lay.layer(l), HP, sm.vec);
    vec_commit<NV, kThreads>(vpf, sm.vec, PredLayerW::vec_count(HP), tid);
    if constexpr (STG) stage_rows(xs0, h, N * LD, wave, lane);
    for (int idx = tid; idx < N * LD; idx += kThreads) {
      agg[idx] = 0.f;
      agg1[idx] = 0.f;
    }
    if (tid < N) {
      sm.pmax[tid] = 0u;
      sm.qmax[tid] = 0u;
    }
    float* st = stash + (size_t)l * (3 * N * HP + 4 * N);
    for (int idx = tid; idx < N; idx += kThreads) gstore4((f4*)(st + 3 * N * HP) + idx, *(const f4*)(sm.x + 4 * idx));
    compute_geo(sm, mg, 1.0f, tid, false);  // gcl.py:308-316
    if constexpr (STG) stage_wait();
    else __syncthreads();
    STAMP(ST_STAGE);
    {
      const NodeCtxH cx = keep ? node_ctx_keep<HP>(hctx(), sm.hk, mg.NC) : hctx();
      node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadOne, kAheadAll, FL>(wb, wbe, Lw.A, h, xs0, true, -1, nullptr, nullptr, Lw.b1, p, nullptr, nullptr, mg.NC, wave,
                                                    lane, tw, cx, pf, Lw.Bm, nullptr, sm.pmax);
      node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadAll, kAheadOne, FL>(wb, wbe, Lw.Bm, h, xs0, false, -1, nullptr, nullptr, nullptr, q, nullptr, nullptr, mg.NC,
                                                    wave, lane, tw, cx, pf, Lw.Wn1h, nullptr, sm.qmax);
    }
    STAMP(ST_NODE);
    __syncthreads();
    STAMP(ST_BARRIER);
    if constexpr (RI) er_start<HP>(ring, wbe, Lw.W2, wave, lane);
    // P, Q -> stash as whole rows (storing them from the accumulators in the GEMM epilogue instead -- 64-byte pieces per
    // lane group -- measured 1.5 % slower on C3)
    for (int idx = tid; idx < N * (HP / 4); idx += kThreads) {
      const int n = idx / (HP / 4), f = 4 * (idx % (HP / 4));
      nstash_store((f4*)st + idx, *(const f4*)(p + n * LD + f));
      nstash_store((f4*)(st + N * HP) + idx, *(const f4*)(q + n * LD + f));
    }
    STAMP(ST_STASH);
    // Rounds of eight 16-slot tiles.  The kernels that are not MR run exactly one and must compile to what they were before
    // rounds existed (a loop that folds away, or a lambda called once, cost C3 1-4 % through hipcc's register allocation):
    // the round is a plain block, and only an MR kernel has the backward jump that repeats it.
    int rd = 0;
    {
    pred_fwd_round:
      const bool more = MR && rd + 1 < mg.rounds;
      const TileCols tc = load_tile(mg, MR ? rd : 0, wave, c);
      const f4 gg = sm.geo[tc.slot];
      f4 acc[T];
      const float d0f = sm.d0[tc.slot];
      const float ub = __builtin_bit_cast(float, sm.pmax[tc.i]) + __builtin_bit_cast(float, sm.qmax[tc.j]) + sm.vec[9 * HP + 1] * gg[0] +
                       sm.vec[9 * HP + 2] * fabsf(d0f);  // (max |c_r|, max |c_d| from the host, behind ba)
      er_gemm_pq<HP>(acc, ring, wbe, Lw.W2, last ? (more ? Lw.W2 : -1) : Lw.Wc1, Lw.b2, Lw.cr, Lw.cd, p + tc.i * LD + 4 * g,
                       q + tc.j * LD + 4 * g, gg[0], d0f, ub, tc.active, wave, lane STAMP_ARGS);
      STAMP(ST_EDGE);
      const int tile = tc.slot >> 4;
      if (tc.active) {
        {  // v (pre-activation of m) -> edge stash for the reverse pass
          f4* sv = (f4*)(estash + edge_stash_off8(l, tile, 0, S, HP)) + lane;
#pragma unroll
          for (int t = 0; t < T; ++t) stash_store(sv + t * 64, acc[t]);
        }
        float sdot = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const f4 m = silu4v(acc[t]);
          acc[t] = m;
          sdot += dot4(m, *(const f4*)(Lw.wa + 16 * t + 4 * g));
        }
        float a = 1.f;
        if (W.attention) a = sigmoid_f(reduce_groups(sdot) + Lw.ba);
        if (g == 0) gstore(astash + (size_t)l * S + tc.slot, a);
        const float sc = a * tc.mk;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = acc[t] * sc;  // e_ij (gcl.py:231-237)
        scatter_runs<HP>(acc, tc, agg, agg1, g);
      }
      STAMP(ST_EDGE_EPI);
      if (!last) {  // coord_model (gcl.py:252-278): trans = dhat * tanh(wc2 . silu(Wc1 e + bc1)) * R * mask
        f4 cp[T];
        er_gemm_regs<HP>(cp, acc, ring, wbe, Lw.Wc1, more ? Lw.W2 : (RI ? -1 : lay.layer(l + 1) + 2 * HP * HP), Lw.bc1, nullptr,
                         tc.active, wave, lane);
        STAMP(ST_EDGE);
        if (tc.active) {
          f4* sc = (f4*)(estash + edge_stash_off8(l, tile, 1, S, HP)) + lane;
          float sdot = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f4 sg = sigmoid4v(cp[t]);  // silu and silu' from one sigmoid, four values per packed instruction
            const f4 ds = sg * (splat(1.0f) + cp[t] * (splat(1.0f) - sg)), sl = cp[t] * sg;
            stash_store(sc + t * 64, ds);
            sdot += dot4(sl, *(const f4*)(Lw.wc2 + 16 * t + 4 * g));
          }
          const float phi = reduce_groups(sdot);
          if (g == 0) gstore(pstash + (size_t)l * S + tc.slot, phi);
          const float tau = (W.use_tanh ? tanhf(phi) * W.coords_range_layer : phi) * tc.mk;
          if (g == 0) *(f4*)(sm.trans + 4 * tc.slot) = (f4){gg[1] * tau, gg[2] * tau, gg[3] * tau, 0.f};
        }
        STAMP(ST_EDGE_EPI);
      }
      if constexpr (MR) {
        if (++rd < mg.rounds) goto pred_fwd_round;
      }
    }
    __syncthreads();
    STAMP(ST_BARRIER);
    if constexpr (STG) stage_rows(xs0, h, N * LD, wave, lane);
    for (int idx = tid; idx < N * (HP / 4); idx += kThreads) {  // agg = partial 0 + partial 1
      const int n = idx / (HP / 4), f = 4 * (idx % (HP / 4));
      if constexpr (STG)  // straight into the staged copy (only this GEMM reads agg)
        *(f4*)(xs1 + n * LD + f) = *(const f4*)(agg + n * LD + f) + *(const f4*)(agg1 + n * LD + f);
      else
        *(f4*)(agg + n * LD + f) = *(const f4*)(agg + n * LD + f) + *(const f4*)(agg1 + n * LD + f);
    }
    if constexpr (STG) stage_wait();
    else __syncthreads();
    STAMP(ST_MISC);
    node_gemm_x<HP, EPI_SILU, true, GN, NH, kAheadOne, kAheadOne, FL>(wb, wbe, Lw.Wn1h, h, xs0, !keep, Lw.Wn1a, agg, xs1, Lw.bn1, p, nullptr, nullptr, mg.NC, wave, lane,
                                            tw, keep ? node_ctx_keep<HP>(hctx(), sm.hk, mg.NC) : hctx(), pf, Lw.Wn2, st + 2 * N * HP /* npre -> stash */);
    STAMP(ST_NODE);
    __syncthreads();
    STAMP(ST_BARRIER);
    vec_prefetch<NV, kThreads>(vpf, wb, PredLayerW::vec_off(lay.layer(l + 1 < W.L ? l + 1 : l), HP), PredLayerW::vec_count(HP), tid);
    if constexpr (STG) {
      stage_rows(xs0, p, N * LD, wave, lane);
      stage_wait();
    }
    node_gemm_x<HP, EPI_RESIDUAL_MASK, false, GN, NH, kAheadOne, kAheadOne, FL>(wb, wbe, Lw.Wn2, p, xs0, true, -1, nullptr, nullptr, Lw.bn2, h, h, mg.mask, mg.NC, wave,
                                                     lane, tw, hctx(), pf, l + 1 < W.L ? lay.layer(l + 1) : -1);
    if (!last) coord_update(sm, mg, 1.0f, tid);
    STAMP(ST_NODE);
    __syncthreads();
    STAMP(ST_BARRIER);
  }
  // readout: mean over the PADDED node count of (embedding_out(h) * mask)   (models.py:553-557, :457)
  {
    const float* ow = w + lay.out_w();
    const float* ob = w + lay.out_b();
    const float* msk = mg.mask;
    small_dots<kThreads>(
        N * K, HP, tid, [=](int qq, int f) { return gload(ow + (qq % K) * HP + f); },
        [=](int qq, int f) { return h[(qq / K) * LD + f]; },
        [=](int qq, float acc) { p[qq] = (acc + gload(ob + qq % K)) * msk[qq / K]; });
    __syncthreads();
    if (tid < K) {
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += p[n * K + tid];
      sm.pred[tid] = s / readout_div;
    }
    __syncthreads();
  }
  STAMP(ST_X0);
}

// ---------------------------------------------------------------------------------------------
// reverse pass: sGrad[N][D] = d( sum_k dpred[k] * pred[k] ) / dz      (dpred in sm.pred[16..16+K))
// buffer roles per layer:  B0 = b0: dagg -> publish buffer    B1 = b1: Q (from the stash) -> publish buffer
//                          B2 = b2: P (stash) -> dP            B3 = b3: dh (running)
//                          B4 = b4: npre (stash) -> dnpre -> dQ
// pub_ch = 16-feature tiles of du published per chunk into [b0 | b1 | pub] (row = 16 pub_ch + 4 floats per slot)
// ---------------------------------------------------------------------------------------------
template <int HP, int SP = 0, bool MR = false, int GN = 0, bool FL = false, class SM = PredSmem<HP, SP, GN>>
__device__ __forceinline__ void pred_backward(const PredDev& W, const MolGraph& mg, const SM& sm, const float* stash,
                                              float* sGrad, float readout_div, int pub_ch, int tid STAMP_DECL, const float* sZin = nullptr) {
  constexpr int LD = HP + 4;
  constexpr int T = HP / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int N = mg.N, D = mg.D, F = W.F, F1 = F + 1, K = W.K, S = mg.S;
  PredLayout lay{HP, F1, K, W.L};
  const float* __restrict__ w = W.w;
  const WBuf wb = make_wbuf(W.w, W.w_bytes);
  const WBuf wbe = SP ? make_wbuf(W.ws, W.ws_bytes) : wb;  // edge-GEMM matrices (w8_split.h)
  const bool tw = W.ktail != 0;                            // H % 16 == 4: the node GEMMs' tail tile (w8_common.h: tail_lane)
  float *B0 = sm.b0, *B1 = sm.b1, *B2 = sm.b2, *dh = sm.b3, *B4 = sm.b4;
  const int tid_ = tid;  // (phases shadow tid with fresh(tid_): device_common.h)
  float* pub = sm.publish();  // [slots][16 pub_ch + 4]
  const int PLD = 16 * pub_ch + 4;
  const float* estash = stash + pred_stash_node_floats(N, HP, W.L);
  const float* astash = estash + (size_t)W.L * S * HP * 2;
  const float* pstash = astash + (size_t)W.L * S;
  const float* dpred = sm.pred + 16;
  const int nslots = mg.ntiles * 16;
  f4* dus = (f4*)const_cast<float*>(pstash + (size_t)W.L * S);  // [S / 16 tiles][T][64] float4: du of every tile (more than one round only)

  if (sZin != nullptr) {
    // split-step mode: the forward ran in an earlier launch; rebuild the two input-geometry tensors it left in LDS
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      const float v = sZin[n * D + d] * mg.mask[n];
      sm.x[4 * n + d] = v;
      sm.x0[4 * n + d] = v;
    }
    __syncthreads();
    compute_geo(sm, mg, 0.f, tid, true);
    __syncthreads();
  }
  // readout backward: dh = mask * (dpred / N_pad) . W_out
  {
    const float* ow = w + lay.out_w();
    for (int idx = tid; idx < N * HP; idx += kThreads) {
      const int n = idx / HP, f = idx % HP;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += (dpred[k] / readout_div) * gload(ow + k * HP + f);
      dh[n * LD + f] = acc * mg.mask[n];
    }
    for (int idx = tid; idx < N * 4; idx += kThreads) sm.dx[idx] = 0.f;
    for (int idx = tid; idx < S; idx += kThreads) sm.dd0[idx] = 0.f;
  }
  __syncthreads();

  typename EdgeRing<HP, SP>::type ring;
  er_init<HP>(ring, sm.ring, W.ktail != 0, W.ws, W.hinv);
  // NH / RI / STG: see edm_forward (w8_edm.h)
  constexpr bool NH = NodeMath<SP>::kF16;
  constexpr bool RI = node_ring_idle(HP, SP, GN);
  constexpr bool STG = GN && !NH;
  float* const xs0 = sm.ring;
  float* const xs1 = sm.ring + stage_stride(N * LD);
  if constexpr (!RI) {
    const int L0 = lay.layer(W.L - 1);
    er_start<HP>(ring, wbe, L0 + 9 * HP * HP /* W2^T of the last layer (its coordinate branch is skipped) */, wave, lane);
  }
  typename NodePFSel<HP, NH>::type pf;
  node_prefetch_x<HP, NH, kAheadOne>(pf, wb, wbe, lay.layer(W.L - 1) + 13 * HP * HP /* Wn2^T of the last layer */, mg.NC, wave, lane, tw);
  auto hctx = [&]() {
    if constexpr (NH) {
      if constexpr (RI) return node_ctx_h<HP>(sm.ring, EdgeRing<HP, SP>::kFloats, mg.NC, W.hinv, tw, sm.hsc);
      else return node_ctx_h<HP>(ring.slot(ring.par ^ 1), EdgeRing<HP, SP>::kFloats / 2, mg.NC, W.hinv, tw, sm.hsc);
    } else {
      return NodeCtxH{1.f, nullptr, nullptr, tw, nullptr};
    }
  };
  constexpr int NV = (PredLayerW::vec_count(HP) + kThreads - 1) / kThreads;
  VecPF<NV> vpf;
  vec_prefetch<NV, kThreads>(vpf, wb, PredLayerW::vec_off(lay.layer(W.L - 1), HP), PredLayerW::vec_count(HP), tid);
  STAMP(ST_X1);
  for (int l = W.L - 1; l >= 0; --l) {
    const bool last = l == W.L - 1;
    const PredLayerW Lw(w, lay.layer(l), HP, sm.vec);
    const float* st = stash + (size_t)l * (3 * N * HP + 4 * N);
    // (a) reload P -> B2, Q -> B1, npre -> B4, x_l ; mask the incoming gradients (h' = (..)*mask, x' = (..)*mask)
    // Every stash read of a thread's first two rounds (all of them up to 19 nodes at HP = 208) and its x row are in flight
    // before the first is used: the stash of a whole batch lives in HBM, and one round trip after another (a loop of
    // load - use, then the x rows) cost three HBM latencies per layer -- 4.2 % of a C3 step (profiles/r05d_stamps_*).
    // Loads of a round that does not exist read the last element instead (no branch around a load: lesson of w8_common.h).
    {
      const int tid = fresh(tid_);
      const int n4 = N * (HP / 4);
#ifdef GAUDI_DIAG_HOT_STASH  // (timing experiment only: wrong results)
      const f4* sp = (const f4*)W.w;
      const f4* sq = (const f4*)(W.w + N * HP);
      const f4* sn = (const f4*)(W.w + 2 * N * HP);
#else
      const f4* sp = (const f4*)st;
      const f4* sq = (const f4*)(st + N * HP);
      const f4* sn = (const f4*)(st + 2 * N * HP);
#endif
      const int i0 = tid < n4 ? tid : n4 - 1, i1 = tid + kThreads < n4 ? tid + kThreads : n4 - 1;
      const f4 xv = gload4((const f4*)(st + 3 * N * HP) + (tid < N ? tid : N - 1));
      const f4 p0 = nstash_load(sp + i0), q0 = nstash_load(sq + i0), v0 = nstash_load(sn + i0);
      const f4 p1 = nstash_load(sp + i1), q1 = nstash_load(sq + i1), v1 = nstash_load(sn + i1);
      vec_commit<NV, kThreads>(vpf, sm.vec, PredLayerW::vec_count(HP), tid);  // (behind the stash reads: both waits overlap)
      STAMP(ST_GUIDE);
      auto put = [&](int idx, f4 pv, f4 qv, f4 nv) {
        const int n = idx / (HP / 4), f = 4 * (idx % (HP / 4));
        *(f4*)(B2 + n * LD + f) = pv;
        *(f4*)(B1 + n * LD + f) = qv;
        *(f4*)(B4 + n * LD + f) = nv;
        *(f4*)(dh + n * LD + f) = *(const f4*)(dh + n * LD + f) * mg.mask[n];
      };
      if (tid < N) {
        *(f4*)(sm.x + 4 * tid) = xv;
        *(f4*)(sm.dx + 4 * tid) = *(const f4*)(sm.dx + 4 * tid) * mg.mask[tid];
      }
      if (tid < n4) put(tid, p0, q0, v0);
      if (tid + kThreads < n4) put(tid + kThreads, p1, q1, v1);
      for (int idx = tid + 2 * kThreads; idx < n4; idx += kThreads) put(idx, nstash_load(sp + idx), nstash_load(sq + idx), nstash_load(sn + idx));
      for (int idx = tid + kThreads; idx < N; idx += kThreads) {
        *(f4*)(sm.x + 4 * idx) = gload4((const f4*)(st + 3 * N * HP) + idx);
        *(f4*)(sm.dx + 4 * idx) = *(const f4*)(sm.dx + 4 * idx) * mg.mask[idx];
      }
    }
    STAMP(ST_PRED_IO);
    __syncthreads();
    if constexpr (STG) stage_rows(xs0, dh, N * LD, wave, lane);
    compute_geo(sm, mg, 1.0f, tid, false);
    if constexpr (STG) stage_wait();
    STAMP(ST_BWD_EDGE);
    // (c) dnpre = (Wn2^T dh) * silu'(npre)  (in place in B4)
    // (GN with fp32 node GEMMs: dnpre feeds (d) only -- it is written straight into the second staging area, not to B4 and back)
    node_gemm_x<HP, EPI_MUL_DSILU, false, GN, NH, kAheadOne, kAheadOne, FL>(wb, wbe, Lw.Wn2t, dh, xs0, true, -1, nullptr, nullptr, nullptr, STG ? xs1 : B4, B4, nullptr,
                                                 mg.NC, wave, lane, tw, hctx(), pf, Lw.Wn1ht);
    __syncthreads();
    // (d) dh += Wn1h^T dnpre ; dagg = Wn1a^T dnpre -> B0 (h is dead)
    {
      const NodeCtxH cx = hctx();
      node_gemm_x<HP, EPI_ACCUM, false, GN, NH, kAheadOne, kAheadAll, FL>(wb, wbe, Lw.Wn1ht, B4, xs1, true, -1, nullptr, nullptr, nullptr, dh, dh, nullptr, mg.NC,
                                                     wave, lane, tw, cx, pf, Lw.Wn1at);
      node_gemm_x<HP, EPI_NONE, false, GN, NH, kAheadAll, 0, FL>(wb, wbe, Lw.Wn1at, B4, xs1, false, -1, nullptr, nullptr, nullptr, B0, nullptr, nullptr,
                                                    mg.NC, wave, lane, tw, cx, pf);
    }
    __syncthreads();
    STAMP(ST_BWD_NODE);
    if constexpr (RI) er_start<HP>(ring, wbe, last ? Lw.W2t : Lw.Wc1t, wave, lane);
    // (e) edge pass: MLP chain backward for the wave's tile, then du of all slots is published CH feature tiles at a time
    //     and every thread sums one (node, 4 features) of dP_i = sum_j du_ij (receiver runs) and dQ_j = sum_i du_ij
    //     (sender lists) in slot order -- no atomics, fixed order
    {
      // One round of eight tiles (every molecule of the reference's datasets; the kernels that are not MR): du stays in
      // registers until it is published.  More rounds (MR kernels): a wave's registers hold one tile, and the publish buffer
      // overwrites dagg / Q, which the later rounds' chains still read -- so every round but the last parks its du in the stash
      // and the publish phase reads it back; the last round's du is published from registers.
      constexpr bool mr = MR;
      int rd = 0;
    pred_bwd_round:  // (only an MR kernel jumps back here: see pred_forward)
      const bool more = MR && rd + 1 < mg.rounds;
      const TileCols tc = load_tile(mg, MR ? rd : 0, wave, c);
      const int tile = tc.slot >> 4;
      const f4 gg = sm.geo[tc.slot];
      const float d0v = sm.d0[tc.slot];
      f4 du[T];
      float a = 0.f, tau = 0.f, dtx = 0.f, dty = 0.f, dtz = 0.f;
      {
        f4 de[T];
        if (tc.active) {
          a = gload(astash + (size_t)l * S + tc.slot);
          dtx = sm.dx[4 * tc.i + 0];  // dtrans = dx'_i
          dty = sm.dx[4 * tc.i + 1];
          dtz = sm.dx[4 * tc.i + 2];
        }
        if (!last) {
          f4 cp[T];
          if (tc.active) {
#ifdef GAUDI_DIAG_HOT_STASH
            const f4* sc = (const f4*)(W.w + (size_t)tile * (T * 256)) + lane;
#else
            const f4* sc = (const f4*)(estash + edge_stash_off8(l, tile, 1, S, HP)) + lane;
#endif
#pragma unroll
            for (int t = 0; t < T; ++t) cp[t] = stash_load(sc + t * 64);  // silu'(cpre)
            const float phi = gload(pstash + (size_t)l * S + tc.slot);
            const float th = tanhf(phi);
            tau = W.use_tanh ? th * W.coords_range_layer : phi;
            const float dtau = (dtx * gg[1] + dty * gg[2] + dtz * gg[3]) * tc.mk;
            const float dphi = W.use_tanh ? dtau * W.coords_range_layer * (1.0f - th * th) : dtau;
#pragma unroll
            for (int t = 0; t < T; ++t)  // dcpre = dphi * wc2 * silu'(cpre)
              cp[t] = *(const f4*)(Lw.wc2 + 16 * t + 4 * g) * dphi * cp[t];
          } else {
#pragma unroll
            for (int t = 0; t < T; ++t) cp[t] = splat(0.f);
          }
          STAMP(ST_B_DCP);
          er_gemm_regs<HP>(de, cp, ring, wbe, Lw.Wc1t, Lw.W2t, nullptr, B0 + tc.i * LD /* + dagg_i (agg_i = sum_j e_ij) */,
                             tc.active, wave, lane);
          STAMP(ST_B_DE);
        } else {
#pragma unroll
          for (int t = 0; t < T; ++t) de[t] = *(const f4*)(B0 + tc.i * LD + 16 * t + 4 * g);
        }
        if (tc.active) {
          // e = m * a * mask ; a = sigmoid(wa . m + ba)
          f4 ve[T];
#ifdef GAUDI_DIAG_HOT_STASH
          const f4* sv = (const f4*)(W.w + (size_t)(8 + tile) * (T * 256)) + lane;
#else
          const f4* sv = (const f4*)(estash + edge_stash_off8(l, tile, 0, S, HP)) + lane;
#endif
#pragma unroll
          for (int t = 0; t < T; ++t) ve[t] = stash_load(sv + t * 64);
          float dadot = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) dadot += dot4(de[t], silu4v(ve[t]));
          const float da = reduce_groups(dadot) * tc.mk;
          const float ds = W.attention ? da * a * (1.0f - a) : 0.f;
          const float am = a * tc.mk;
#pragma unroll
          for (int t = 0; t < T; ++t)  // dv = (de*a*mask + ds*wa) * silu'(v)
            de[t] = (de[t] * am + *(const f4*)(Lw.wa + 16 * t + 4 * g) * ds) * dsilu4v(ve[t]);
        }
        STAMP(ST_B_DV);
        // next edge GEMM of the chain: Wc1^T of layer l-1 (W2^T when that layer is ... never the last), none after layer 0
        // split form: nothing is prefetched across the publish phase (the ring is part of the publish buffer there); the
        // next layer's first group is requested right after it instead
        er_gemm_regs<HP>(du, de, ring, wbe, Lw.W2t,
                         more ? (last ? Lw.W2t : Lw.Wc1t) : ((SP == 0 && l > 0) ? lay.layer(l - 1) + 10 * HP * HP : -1), nullptr, nullptr,
                         tc.active, wave, lane);  // dt1
        STAMP(ST_B_DT1);
      }
      if (tc.active) {
        // du = dt1 * silu'(u) ; dr = cr . du ; dd0 = cd . du
        const float* pp = B2 + tc.i * LD + 4 * g;
        const float* qq = B1 + tc.j * LD + 4 * g;
        float drdot = 0.f, dd0dot = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const f4 u = edge_u(pp, qq, Lw.cr, Lw.cd, g, t, gg[0], d0v);
          du[t] = du[t] * dsilu4v(u);
          drdot += dot4(du[t], *(const f4*)(Lw.cr + 16 * t + 4 * g));
          dd0dot += dot4(du[t], *(const f4*)(Lw.cd + 16 * t + 4 * g));
        }
        const float dr = reduce_groups(drdot), dd0v = reduce_groups(dd0dot);
        if (g == 0) {
          // d/d(diff) of r = |diff|^2 and dhat = diff / (sqrt(r + 1e-8) + 1)   (gcl.py:308-316)
          const int i = tc.i, j = tc.j;
          const float fx = sm.x[4 * i + 0] - sm.x[4 * j + 0];
          const float fy = sm.x[4 * i + 1] - sm.x[4 * j + 1];
          const float fz = sm.x[4 * i + 2] - sm.x[4 * j + 2];
          const float nrm = sqrtf(gg[0] + 1e-8f), den = nrm + 1.0f;
          const float cx = dtx * tau * tc.mk, cy = dty * tau * tc.mk, cz = dtz * tau * tc.mk;
          const float k1 = (cx * fx + cy * fy + cz * fz) / (den * den * nrm);
          *(f4*)(sm.trans + 4 * tc.slot) = (f4){cx / den - fx * k1 + 2.0f * fx * dr, cy / den - fy * k1 + 2.0f * fy * dr,
                                                cz / den - fz * k1 + 2.0f * fz * dr, 0.f};
          sm.dd0[tc.slot] += dd0v;
        }
      }
      if constexpr (MR) {
        if (more) {  // (the last round's du stays in registers until it is published)
          if (tc.active) {
            f4* sd = dus + (size_t)tile * (T * 64) + lane;
#pragma unroll
            for (int t = 0; t < T; ++t) stash_store(sd + t * 64, du[t]);
          }
          ++rd;
          goto pred_bwd_round;
        }
      }
      STAMP(ST_B_DU);
      __syncthreads();  // every wave is done with P (B2), Q (B1) and dagg (B0): the publish buffer may overwrite B0 / B1
      STAMP(ST_BWD_BARRIER);
      for (int t0 = 0; t0 < T; t0 += pub_ch) {
        const int t1 = t0 + pub_ch < T ? t0 + pub_ch : T;
        if constexpr (!mr) {
          if (tc.active) {
            float* row = pub + tc.slot * PLD + 4 * g - 16 * t0;
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (t >= t0 && t < t1) *(f4*)(row + 16 * t) = du[t];
          }
        } else {
          for (int r2 = 0; r2 + 1 < mg.rounds; ++r2) {  // the earlier rounds' tiles: parked in the stash
            const TileCols tp = load_tile(mg, r2, wave, c);
            if (tp.active) {
              float* row = pub + tp.slot * PLD + 4 * g - 16 * t0;
              const f4* sd = dus + (size_t)(tp.slot >> 4) * (T * 64) + lane;
#pragma unroll
              for (int t = 0; t < T; ++t)
                if (t >= t0 && t < t1) *(f4*)(row + 16 * t) = stash_load(sd + t * 64);
            }
          }
          if (tc.active) {  // the last round's tile: still in registers
            float* row = pub + tc.slot * PLD + 4 * g - 16 * t0;
#pragma unroll
            for (int t = 0; t < T; ++t)
              if (t >= t0 && t < t1) *(f4*)(row + 16 * t) = du[t];
          }
        }
        __syncthreads();
        STAMP(ST_B_V);
        const int nf4 = (t1 - t0) * 4;  // float4 per slot in this chunk
        for (int idx = tid; idx < N * nf4; idx += kThreads) {
          const int n = idx / nf4, f = 4 * (idx % nf4);
          const uint32_t sg = mg.seg[n];
          const int rs = sg >> 16, rl = sg & 0xffff;
          f4 sp = splat(0.f), sq = splat(0.f);
          for (int k = 0; k < rl; ++k) sp += *(const f4*)(pub + (rs + k) * PLD + f);
          const int s0 = mg.soff[n], s1 = mg.soff[n + 1];
          for (int k = s0; k < s1; ++k) sq += *(const f4*)(pub + (int)mg.sidx[k] * PLD + f);
          *(f4*)(B2 + n * LD + 16 * t0 + f) = sp;  // dP_n
          *(f4*)(B4 + n * LD + 16 * t0 + f) = sq;  // dQ_n
        }
        STAMP(ST_B_EV);
        __syncthreads();
        STAMP(ST_B_CP);
      }
      if (SP != 0 && !RI && l > 0)
        er_start<HP>(ring, wbe, lay.layer(l - 1) + 10 * HP * HP /* Wc1^T of the layer below */, wave, lane);
      node_prefetch_x<HP, NH, kAheadOne>(pf, wb, wbe, Lw.At, mg.NC, wave, lane, tw);
      STAMP(ST_BWD_COL);
    }
    // dx <- dx*mask + sum_{e: i=n} ddiff_e - sum_{e: j=n} ddiff_e, in slot order
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      const uint32_t sg = mg.seg[n];
      const int rs = sg >> 16, rl = sg & 0xffff;
      float acc = sm.dx[4 * n + d];
      for (int k = 0; k < rl; ++k) acc += sm.trans[4 * (rs + k) + d];
      const int s0 = mg.soff[n], s1 = mg.soff[n + 1];
      for (int k = s0; k < s1; ++k) acc -= sm.trans[4 * (int)mg.sidx[k] + d];
      sm.dx[4 * n + d] = acc;
    }
    // (f) dh += A^T dP + Bm^T dQ
    vec_prefetch<NV, kThreads>(vpf, wb, PredLayerW::vec_off(lay.layer(l > 0 ? l - 1 : 0), HP), PredLayerW::vec_count(HP), tid);
    if constexpr (STG) {  // (the publish loop ended on a barrier: dP / dQ are complete, the ring is free)
      stage_rows(xs0, B2, N * LD, wave, lane);
      stage_rows(xs1, B4, N * LD, wave, lane);
      stage_wait();
    }
    node_gemm_x<HP, EPI_ACCUM, true, GN, NH, kAheadOne, kAheadOne, FL>(wb, wbe, Lw.At, B2, xs0, true, Lw.Bmt, B4, xs1, nullptr, dh, dh, nullptr, mg.NC, wave, lane, tw,
                                             hctx(), pf, l > 0 ? lay.layer(l - 1) + 13 * HP * HP : -1);
    __syncthreads();
    STAMP(ST_BWD_NODE);
  }
  (void)nslots;

  // embedding backward (time column dropped), d0 backward, input masking
  {
    const float* ew = w + lay.emb_w();
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      const uint32_t sg = mg.seg[n];
      const int rs = sg >> 16, rl = sg & 0xffff;
      float acc = sm.dx[4 * n + d];
      for (int k = 0; k < rl; ++k) {
        const int s = rs + k, j = ew_j(mg.edge[s]);
        acc += 2.0f * (sm.x0[4 * n + d] - sm.x0[4 * j + d]) * sm.dd0[s];
      }
      const int s0 = mg.soff[n], s1 = mg.soff[n + 1];
      for (int k = s0; k < s1; ++k) {
        const int s = mg.sidx[k], i = ew_i(mg.edge[s]);
        acc -= 2.0f * (sm.x0[4 * i + d] - sm.x0[4 * n + d]) * sm.dd0[s];
      }
      sGrad[n * D + d] = acc * mg.mask[n];
    }
    const float* msk = mg.mask;
    small_dots<kThreads>(
        N * F, HP, tid, [=](int qq, int f) { return gload(ew + f * F1 + qq % F); },
        [=](int qq, int f) { return dh[(qq / F) * LD + f]; },
        [=](int qq, float acc) { sGrad[(qq / F) * D + 3 + qq % F] = acc * msk[qq / F]; });
  }
  __syncthreads();
}

// unit-test entry: pred (and optionally grad into sGrad) for z in sZ
template <int HP, int SP = 0, bool MR = false, int GN = 0, bool PG = false>
__device__ __forceinline__ void predictor_entry(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad,
                                                float* sTmp, float* sMean, float t_val, const float* dpred, bool want_grad,
                                                float* pred_out, float readout_div, float* stash, int pubx, int pub_ch, int tid STAMP_DECL,
                                                float* gnode = nullptr) {
  (void)sTmp; (void)sMean;
  PredSmem<HP, SP, GN, PG> sm;
  sm.carve(net, mg.N, mg.S, pubx, gnode);
  pred_forward<HP, SP, MR, GN>(W, mg, sm, sZ, t_val, stash, readout_div, tid STAMP_ARGS);
  if (tid < W.K) {
    if (pred_out) pred_out[tid] = sm.pred[tid];
    sm.pred[16 + tid] = dpred ? dpred[tid] : 0.f;
  }
  __syncthreads();
  if (want_grad) pred_backward<HP, SP, MR, GN>(W, mg, sm, stash, sGrad, readout_div, pub_ch, tid STAMP_ARGS);
}

template <class SM>
__device__ __forceinline__ void guidance_seed(const PredDev& W, const SM& sm, const float* target_w, float scale,
                                              float* pred_out, int tid, int phase, const float* dpred_ext);
__device__ __forceinline__ void guidance_apply(const MolGraph& mg, float* sZ, float* sGrad, float* sMean, float sigma, int tid);

// guidance of one reverse step (en_diffusion.py:899-920): z_s <- z_s - sigma * P(clip(grad))
// phase 0: fused (target linear in pred: dT/dpred = target_w);  phase 1: predictor forward only, pred -> pred_out
// (the host evaluates an arbitrary target on it);  phase 2: reverse pass + update with dT/dpred = dpred_ext.
template <int HP, int SP = 0, bool MR = false, int GN = 0, bool PG = false>
__device__ __forceinline__ void guidance_update(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad,
                                                float* sTmp, float* sMean, float t_val, float sigma, const float* target_w,
                                                float scale, float* pred_out, float readout_div, float* stash, int pubx,
                                                int pub_ch, int tid STAMP_DECL, int phase, const float* dpred_ext,
                                                float* gnode = nullptr) {
  (void)sTmp;
  const int N = mg.N, D = mg.D;
  PredSmem<HP, SP, GN, PG> sm;
  sm.carve(net, N, mg.S, pubx, gnode);
  sm.hk = lds_at(mg.hk);
  if (phase != 2) pred_forward<HP, SP, MR, GN>(W, mg, sm, sZ, t_val, stash, readout_div, tid STAMP_ARGS);
  guidance_seed(W, sm, target_w, scale, pred_out, tid, phase, dpred_ext);
  if (phase == 1) return;
  pred_backward<HP, SP, MR, GN>(W, mg, sm, stash, sGrad, readout_div, pub_ch, tid STAMP_ARGS, phase == 2 ? sZ : nullptr);
  guidance_apply(mg, sZ, sGrad, sMean, sigma, tid);
}

// pred -> pred_out (split mode) and the seed of the reverse pass: d(energy)/dpred = scale * dT/dpred
template <class SM>
__device__ __forceinline__ void guidance_seed(const PredDev& W, const SM& sm, const float* target_w, float scale,
                                              float* pred_out, int tid, int phase, const float* dpred_ext) {
  if (tid < W.K) {
    if (pred_out && phase != 2) pred_out[tid] = sm.pred[tid];
    sm.pred[16 + tid] = (phase == 2 ? dpred_ext[tid] : target_w[tid]) * scale;
  }
  __syncthreads();
}

// clip, project, apply (en_diffusion.py:905-920), per molecule = per component of a packed graph.  A component's sums
// visit its elements in the order (and on the lanes) the molecule's own workgroup would: element index = (node inside the
// molecule) * D + column, so the result does not depend on which slots of which workgroup hold the molecule.
__device__ __forceinline__ void guidance_apply(const MolGraph& mg, float* sZ, float* sGrad, float* sMean, float sigma, int tid) {
  const int N = mg.N, D = mg.D;
  // clip_coef = min(1, 10 / (||g||_2 + 1e-6)) over all N*(3+F) entries   (en_diffusion.py:905-909)
  if (tid < 64 * mg.ncomp) {
    const int k = tid >> 6, lane = tid & 63;
    float s = 0.f;
    for (int n = 0; n < N; ++n) {
      if (mg_comp(mg, n) != k) continue;
      const int local = ((mg.row[n] & 0x0fffffff) % mg.NR) * D;
      for (int d = 0; d < D; ++d)
        if (((local + d) & 63) == lane) s += sGrad[n * D + d] * sGrad[n * D + d];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sMean[4 * k + 3] = fminf(10.0f / (sqrtf(s) + 1e-6f), 1.0f);
  }
  __syncthreads();
  for (int e = tid; e < N * D; e += kThreads) {
    const int k = mg_comp(mg, e / D);
    sGrad[e] *= k < mg.ncomp ? sMean[4 * k + 3] : 0.f;
  }
  __syncthreads();
  if (tid < 3 * mg.ncomp) {  // masked mean of the x part of the gradient (en_diffusion.py:911-919)
    const int k = tid / 3, d = tid % 3;
    float s = 0.f, cnt = 0.f;
    for (int n = 0; n < N; ++n)
      if (mg_comp(mg, n) == k) { s += sGrad[n * D + d]; cnt += mg.mask[n]; }
    sMean[4 * k + d] = s / fmaxf(cnt, 1.0f);
  }
  __syncthreads();
  for (int e = tid; e < N * D; e += kThreads) {
    const int n = e / D, d = e % D, k = mg_comp(mg, n);
    float gv = sGrad[e];
    if (d < 3 && k < mg.ncomp) gv = gv - sMean[4 * k + d] * mg.mask[n];
    sZ[e] = sZ[e] - sigma * gv;
  }
  __syncthreads();
}

}  // namespace w8
}  // namespace gaudi
