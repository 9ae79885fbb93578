// pred_device.h -- EGNN_predictor (edm/egnn_predictor/models.py:433-457,543-560; gcl.py:225-316):
// forward with a small per-layer activation stash (h, agg, x per node -- edge activations are
// recomputed), the hand-written reverse pass that replaces torch.autograd.grad at
// en_diffusion.py:900-903, and the guidance epilogue (:905-920).  One workgroup = one molecule.
#pragma once
#include "device_common.h"
#include "edm_device.h"

#ifndef GAUDI_BWD_TILES
#define GAUDI_BWD_TILES 1  // 16-edge tiles per wave per round of the reverse edge pass
#endif

namespace gaudi {

// Packed predictor weights (floats), HP = padded hidden, PK = HP*HP:
//   head : emb_w [HP][F1] | emb_b [HP] | out_w [K][HP] | out_b [16]
//   layer: A, Bm, W2, Wc1, Wn1h, Wn1a, Wn2  and their transposes (14 PK)
//          | cr, cd, b1, b2, wa, bc1, wc2, bn1, bn2 (9 HP) | ba (16)
struct PredLayout {
  int HP, F1, K, L;
  __host__ __device__ int pk() const { return HP * HP; }
  __host__ __device__ int emb_w() const { return 0; }
  __host__ __device__ int emb_b() const { return align16(HP * F1); }
  __host__ __device__ int out_w() const { return emb_b() + HP; }
  __host__ __device__ int out_b() const { return out_w() + align16(K * HP); }
  __host__ __device__ int layers() const { return out_b() + 16; }
  __host__ __device__ int layer_size() const { return 14 * pk() + 9 * HP + 16; }
  __host__ __device__ int layer(int l) const { return layers() + l * layer_size(); }
  __host__ __device__ int total() const { return layers() + L * layer_size(); }
};

struct PredDev {
  const float* w;
  unsigned w_bytes;
  int F, K, L, attention, use_tanh;
  float coords_range_layer;  // coords_range / n_layers (egnn_predictor/models.py:515)
  int ktail;                 // as EdmDev::ktail
  const float* ws;           // as EdmDev::ws
  unsigned ws_bytes;
  float hinv;                // as EdmDev::hinv
};

template <int HP, bool GN = false>  // GN: node buffers in a per-molecule global scratch (edm_device.h: NetSmem)
struct PredSmem {
  static constexpr bool kGlobalNodes = GN;
  static constexpr int kEF = 2;  // the predictor has no sin_embedding (egnn_predictor/models.py:452: edge_attr = d0)
  float *b0, *b1, *b2, *b3, *b4;  // [N][HP+4] node buffers (roles change per phase, see below)
  float* scr;                     // [4][16][HP+4]
  float *x, *x0, *dx;             // [N][4]
  f4* geo;                        // [4][EW]
  float *d0, *trans, *dd0;        // [4][EW], [4][EW][4], [4][EW]
  float* pred;                    // [16] pred | [16] dpred
  float* vec;                     // [10*HP] the current layer's vectors (cr,cd,b1,b2,wa,bc1,wc2,bn1,bn2,ba)
  __host__ __device__ static int floats(int N, int EW) {
    return (GN ? 0 : 5 * N * (HP + 4)) + kWaves * 16 * (HP + 4) + 12 * N + kWaves * EW * 10 + 32 + 10 * HP;
  }
  __device__ void carve(float* base, int N, int EW, float* gnode = nullptr) {
    constexpr int LD = HP + 4;
    float*& nb = GN ? gnode : base;
    b0 = nb; nb += N * LD;
    b1 = nb; nb += N * LD;
    b2 = nb; nb += N * LD;
    b3 = nb; nb += N * LD;
    b4 = nb; nb += N * LD;
    scr = base; base += kWaves * 16 * LD;
    x = base; base += 4 * N;
    x0 = base; base += 4 * N;
    dx = base; base += 4 * N;
    geo = (f4*)base; base += kWaves * EW * 4;
    d0 = base; base += kWaves * EW;
    trans = base; base += kWaves * EW * 4;
    dd0 = base; base += kWaves * EW;
    pred = base; base += 32;
    vec = base;
  }
};

__device__ __forceinline__ float dot4(f4 a, f4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// float offsets of one layer's tensors inside the packed weight buffer
struct PredLayerW {
  int A, Bm, W2, Wc1, Wn1h, Wn1a, Wn2, At, Bmt, W2t, Wc1t, Wn1ht, Wn1at, Wn2t;
  int V;  // start of the vector block (9 HP + 16) in the weight buffer
  const float *cr, *cd, *b1, *b2, *wa, *bc1, *wc2, *bn1, *bn2;  // LDS copies (stage() first)
  float ba;
  // copy the vector block to LDS; the caller places a barrier before the first use
  __host__ __device__ static constexpr int vec_off(int L0, int HP) { return L0 + 14 * HP * HP; }
  __host__ __device__ static constexpr int vec_count(int HP) { return 9 * HP + 16; }
  __device__ PredLayerW(const float* w, int L0, int HP, const float* sVec) {
    const int PK = HP * HP;
    A = L0; Bm = L0 + PK; W2 = L0 + 2 * PK; Wc1 = L0 + 3 * PK; Wn1h = L0 + 4 * PK; Wn1a = L0 + 5 * PK;
    Wn2 = L0 + 6 * PK; At = L0 + 7 * PK; Bmt = L0 + 8 * PK; W2t = L0 + 9 * PK; Wc1t = L0 + 10 * PK;
    Wn1ht = L0 + 11 * PK; Wn1at = L0 + 12 * PK; Wn2t = L0 + 13 * PK;
    V = L0 + 14 * PK;
    cr = sVec; cd = sVec + HP; b1 = sVec + 2 * HP; b2 = sVec + 3 * HP; wa = sVec + 4 * HP; bc1 = sVec + 5 * HP;
    wc2 = sVec + 6 * HP; bn1 = sVec + 7 * HP; bn2 = sVec + 8 * HP;
    ba = gload(w + V + 9 * HP);
  }
};

// stash per molecule: node part  L x { P [N][HP] | Q [N][HP] | npre [N][HP] | x [N][4] }   (P = A h + b1, Q = B h,
//                                npre = pre-activation of the node MLP: the reverse pass reloads instead of recomputing)
//                     edge part  L x 4 waves x (EW/16) tiles x { v | cpre } x [HP/16][64 lanes] float4
// (the two edge pre-activations, in the accumulator layout they are produced in: 1 KiB per store instruction).
__host__ __device__ inline long long pred_stash_node_floats(int N, int HP, int L) {
  return (long long)L * (3LL * N * HP + 4LL * N);
}
__host__ __device__ inline long long pred_stash_edge_floats(int HP, int L, int EW) {
  return (long long)L * kWaves * EW * HP * 2;
}
// ... + attention gate a_ij of every edge slot, L x 4 waves x EW floats (the reverse pass needs it before it needs v)
__host__ __device__ inline long long pred_stash_floats(int N, int HP, int L, int EW) {
  return pred_stash_node_floats(N, HP, L) + pred_stash_edge_floats(HP, L, EW) + (long long)L * kWaves * EW;
}
// float offset of tile `tile` of wave `wave` in layer l, array arr (0 = v, 1 = cpre), inside the edge part
__device__ __forceinline__ size_t edge_stash_off(int l, int wave, int tile, int arr, int EW, int HP) {
  return ((((size_t)l * kWaves + wave) * (EW / 16) + tile) * 2 + arr) * (size_t)(16 * HP);
}

// ---------------------------------------------------------------------------------------------
// forward: pred[K] -> sm.pred[0..K)
// buffers: h = b0, P = b1, Q = b2, agg = b3
// ---------------------------------------------------------------------------------------------
template <int HP, class SM = PredSmem<HP>>
__device__ __forceinline__ void pred_forward(const PredDev& W, const MolGraph& mg, const SM& sm, const float* sZ, float t_val,
                             float* stash, float readout_div, int tid STAMP_DECL) {
  constexpr int LD = HP + 4;
  constexpr int T = HP / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int N = mg.N, D = mg.D, F = W.F, F1 = F + 1, K = W.K;
  PredLayout lay{HP, F1, K, W.L};
  const float* __restrict__ w = W.w;
  const WBuf wb = make_wbuf(W.w, W.w_bytes);
  float *h = sm.b0, *p = sm.b1, *q = sm.b2, *agg = sm.b3;
  float* estash = stash + pred_stash_node_floats(N, HP, W.L);
  float* astash = estash + pred_stash_edge_floats(HP, W.L, mg.EW);

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
      for (int k = 0; k < F; ++k) acc += ew[f * F1 + k] * (sZ[n * D + 3 + k] * m);
      acc += ew[f * F1 + F] * t_val;
      h[n * LD + f] = acc + eb[f];
    }
  }
  __syncthreads();
  compute_geo(sm, mg, 0.f, wave, lane, true);  // edge_attr = |x_i - x_j|^2 of the input (models.py:452)
  NodePF<HP> pf;
  node_prefetch<HP>(pf, wb, lay.layer(0), wave, lane);
  constexpr int NV = (PredLayerW::vec_count(HP) + kThreads - 1) / kThreads;
  VecPF<NV> vpf;  // the next layer's vectors, loaded one node GEMM ahead
  vec_prefetch(vpf, wb, PredLayerW::vec_off(lay.layer(0), HP), PredLayerW::vec_count(HP), tid);
  STAMP(ST_PRED_IO);

  for (int l = 0; l < W.L; ++l) {
    const bool last = l == W.L - 1;  // the last layer's coordinate update never reaches the readout
    const PredLayerW Lw(w, lay.layer(l), HP, sm.vec);
    vec_commit(vpf, sm.vec, PredLayerW::vec_count(HP), tid);  // previous readers are behind the barrier that ended layer l-1
    __syncthreads();
    STAMP(ST_STAGE);
    float* st = stash + (size_t)l * (3 * N * HP + 4 * N);
    for (int idx = tid; idx < N; idx += kThreads) ((f4*)(st + 3 * N * HP))[idx] = *(const f4*)(sm.x + 4 * idx);
    compute_geo(sm, mg, 1.0f, wave, lane, false);  // gcl.py:308-316
    STAMP(ST_GEO);
    node_gemm<HP, EPI_NONE, true>(wb, Lw.A, h, -1, nullptr, Lw.b1, p, nullptr, nullptr, mg.NC, wave, lane, &pf, Lw.Bm);
    node_gemm<HP, EPI_NONE, true>(wb, Lw.Bm, h, -1, nullptr, nullptr, q, nullptr, nullptr, mg.NC, wave, lane, &pf);
    for (int idx = tid; idx < N * LD; idx += kThreads) agg[idx] = 0.f;
    STAMP(ST_NODE);
    __syncthreads();
    STAMP(ST_BARRIER);
    for (int idx = tid; idx < N * (HP / 4); idx += kThreads) {  // P, Q -> stash
      const int n = idx / (HP / 4), f = 4 * (idx % (HP / 4));
      ((f4*)st)[idx] = *(const f4*)(p + n * LD + f);
      ((f4*)(st + N * HP))[idx] = *(const f4*)(q + n * LD + f);
    }
    {
      SegSum<HP> ss;
      ss.init();
      float* scr = sm.scr + wave * 16 * LD;
      for (int tp = 0; tp < mg.npairs; ++tp) {
        EdgeCol ec[2];
        float mk2[2];
        f4 geo2[2];
        load_cols<SM, 2>(sm, mg, wave, tp * 32, c, ec, mk2, geo2);
        f4 acc[2][T];
        edge_gemm_from_pq<HP, 2>(acc, wb, Lw.W2, Lw.b2, Lw.cr, Lw.cd, p, q, ec, lane);
        STAMP(ST_EDGE);
#pragma unroll
        for (int e = 0; e < 2; ++e) {  // v (pre-activation of m) -> edge stash for the reverse pass
          f4* sv = (f4*)(estash + edge_stash_off(l, wave, tp * 2 + e, 0, mg.EW, HP)) + lane;
#pragma unroll
          for (int t = 0; t < T; ++t) stash_store(sv + t * 64, acc[e][t]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float sdot = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f4 m = silu4(acc[e][t]);
            acc[e][t] = m;
            sdot += dot4(m, *(const f4*)(Lw.wa + 16 * t + 4 * g));
          }
          float a = 1.f;
          if (W.attention) a = sigmoid_f(reduce_groups(sdot) + Lw.ba);
          if (g == 0) astash[((size_t)l * kWaves + wave) * mg.EW + tp * 32 + e * 16 + c] = a;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            acc[e][t] = acc[e][t] * a * mk2[e];  // e_ij (gcl.py:231-237)
            *(f4*)(scr + c * LD + 16 * t + 4 * g) = acc[e][t];
          }
          wave_lds_fence();
          ss.add_tile(scr, ec[e].i, agg, 1.0f, lane);
          wave_lds_fence();
        }
        if (!last) {  // coord_model (gcl.py:252-278): trans = dhat * tanh(wc2 . silu(Wc1 e + bc1)) * R * mask
          f4 cp[2][T];
          const float* const noinit[2] = {nullptr, nullptr};
          STAMP(ST_EDGE_EPI);
          edge_gemm_from_regs<HP, 2>(cp, acc, wb, Lw.Wc1, Lw.bc1, noinit, lane);
          STAMP(ST_EDGE);
#pragma unroll
          for (int e = 0; e < 2; ++e) {  // cpre -> edge stash
            f4* sc = (f4*)(estash + edge_stash_off(l, wave, tp * 2 + e, 1, mg.EW, HP)) + lane;
#pragma unroll
            for (int t = 0; t < T; ++t) stash_store(sc + t * 64, cp[e][t]);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) sdot += dot4(silu4(cp[e][t]), *(const f4*)(Lw.wc2 + 16 * t + 4 * g));
            const float phi = reduce_groups(sdot);
            const float tau = W.use_tanh ? tanhf(phi) * W.coords_range_layer : phi;
            if (g == 0) {
              const int slot = wave * mg.EW + tp * 32 + e * 16 + c;
              const f4 gg = geo2[e];
              *(f4*)(sm.trans + 4 * slot) = (f4){gg[1] * tau * mk2[e], gg[2] * tau * mk2[e], gg[3] * tau * mk2[e], 0.f};
            }
          }
        }
      }
      node_prefetch<HP>(pf, wb, Lw.Wn1h, wave, lane);
      ss.flush(agg, 1.0f, lane);
    }
    STAMP(ST_EDGE_EPI);
    __syncthreads();
    STAMP(ST_BARRIER);
    node_gemm<HP, EPI_SILU, true>(wb, Lw.Wn1h, h, Lw.Wn1a, agg, Lw.bn1, p, nullptr, nullptr, mg.NC, wave, lane, &pf, Lw.Wn2,
                                  st + 2 * N * HP /* npre -> stash */);
    STAMP(ST_NODE);
    __syncthreads();
    STAMP(ST_BARRIER);
    vec_prefetch(vpf, wb, PredLayerW::vec_off(lay.layer(l + 1 < W.L ? l + 1 : l), HP), PredLayerW::vec_count(HP), tid);
    node_gemm<HP, EPI_RESIDUAL_MASK, true>(wb, Lw.Wn2, p, -1, nullptr, Lw.bn2, h, h, mg.mask, mg.NC, wave, lane, &pf,
                                           l + 1 < W.L ? lay.layer(l + 1) : -1);
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
    small_dots(
        N * K, HP, tid, [=](int q, int f) { return ow[(q % K) * HP + f]; },
        [=](int q, int f) { return h[(q / K) * LD + f]; },
        [=](int q, float acc) { p[q] = (acc + ob[q % K]) * msk[q / K]; });
    __syncthreads();
    if (tid < K) {
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += p[n * K + tid];
      sm.pred[tid] = s / readout_div;
    }
    __syncthreads();
  }
  STAMP(ST_PRED_IO);
}

// ---------------------------------------------------------------------------------------------
// reverse pass: sGrad[N][D] = d( sum_k dpred[k] * pred[k] ) / dz      (dpred in sm.pred[16..16+K))
// buffer roles per layer:  B0 = b0: dagg             B1 = b1: Q (from the stash)
//                          B2 = b2: P (stash) -> dP (in place, rows owned by the wave that reduces them)
//                          B3 = b3: dh (running)     B4 = b4: npre (stash) -> dnpre -> dQ accumulator
// ---------------------------------------------------------------------------------------------
template <int HP, class SM = PredSmem<HP>>
__device__ __forceinline__ void pred_backward(const PredDev& W, const MolGraph& mg, const SM& sm, const float* stash,
                              float* sGrad, float readout_div, int tid STAMP_DECL, const float* sZin = nullptr) {
  constexpr int LD = HP + 4;
  constexpr int T = HP / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int N = mg.N, D = mg.D, F = W.F, F1 = F + 1, K = W.K, EW = mg.EW;
  PredLayout lay{HP, F1, K, W.L};
  const float* __restrict__ w = W.w;
  const WBuf wb = make_wbuf(W.w, W.w_bytes);
  float *B0 = sm.b0, *B1 = sm.b1, *B2 = sm.b2, *dh = sm.b3, *B4 = sm.b4;
  const float* estash = stash + pred_stash_node_floats(N, HP, W.L);
  const float* astash = estash + pred_stash_edge_floats(HP, W.L, EW);
  const float* dpred = sm.pred + 16;
  int ntmax = 0;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) ntmax = max(ntmax, 2 * mg.npairs_all[k]);
  const int nt_me = 2 * mg.npairs;

  if (sZin != nullptr) {
    // split-step mode: the forward ran in an earlier launch; rebuild the two input-geometry tensors it left in LDS
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      const float v = sZin[n * D + d] * mg.mask[n];
      sm.x[4 * n + d] = v;
      sm.x0[4 * n + d] = v;
    }
    __syncthreads();
    compute_geo(sm, mg, 0.f, wave, lane, true);
    __syncthreads();
  }
  // readout backward: dh = mask * (dpred / N_pad) . W_out
  {
    const float* ow = w + lay.out_w();
    for (int idx = tid; idx < N * HP; idx += kThreads) {
      const int n = idx / HP, f = idx % HP;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += (dpred[k] / readout_div) * ow[k * HP + f];
      dh[n * LD + f] = acc * mg.mask[n];
    }
    for (int idx = tid; idx < N * 4; idx += kThreads) sm.dx[idx] = 0.f;
    for (int idx = tid; idx < kWaves * EW; idx += kThreads) sm.dd0[idx] = 0.f;
  }
  __syncthreads();

  NodePF<HP> pf;
  node_prefetch<HP>(pf, wb, lay.layer(W.L - 1) + 13 * HP * HP /* Wn2^T of the last layer */, wave, lane);
  constexpr int NV = (PredLayerW::vec_count(HP) + kThreads - 1) / kThreads;
  VecPF<NV> vpf;
  vec_prefetch(vpf, wb, PredLayerW::vec_off(lay.layer(W.L - 1), HP), PredLayerW::vec_count(HP), tid);
  STAMP(ST_PRED_IO);
  for (int l = W.L - 1; l >= 0; --l) {
    const bool last = l == W.L - 1;
    const PredLayerW Lw(w, lay.layer(l), HP, sm.vec);
    vec_commit(vpf, sm.vec, PredLayerW::vec_count(HP), tid);
    __syncthreads();
    STAMP(ST_STAGE);
    const float* st = stash + (size_t)l * (3 * N * HP + 4 * N);
    // (a) reload P -> B2, Q -> B1, npre -> B4, x_l ; mask the incoming gradients (h' = (..)*mask, x' = (..)*mask)
    for (int idx = tid; idx < N * (HP / 4); idx += kThreads) {
      const int n = idx / (HP / 4), f = 4 * (idx % (HP / 4));
      const f4 pv = ((const f4*)st)[idx], qv = ((const f4*)(st + N * HP))[idx], nv = ((const f4*)(st + 2 * N * HP))[idx];
      *(f4*)(B2 + n * LD + f) = pv;
      *(f4*)(B1 + n * LD + f) = qv;
      *(f4*)(B4 + n * LD + f) = nv;
      *(f4*)(dh + n * LD + f) = *(const f4*)(dh + n * LD + f) * mg.mask[n];
    }
    for (int idx = tid; idx < N; idx += kThreads) {
      *(f4*)(sm.x + 4 * idx) = ((const f4*)(st + 3 * N * HP))[idx];
      *(f4*)(sm.dx + 4 * idx) = *(const f4*)(sm.dx + 4 * idx) * mg.mask[idx];
    }
    __syncthreads();
    STAMP(ST_STASH);
    compute_geo(sm, mg, 1.0f, wave, lane, false);
    STAMP(ST_GEO);
    // (c) dnpre = (Wn2^T dh) * silu'(npre)  (in place in B4)
    node_gemm<HP, EPI_MUL_DSILU, true>(wb, Lw.Wn2t, dh, -1, nullptr, nullptr, B4, B4, nullptr, mg.NC, wave, lane, &pf, Lw.Wn1ht);
    __syncthreads();
    // (d) dh += Wn1h^T dnpre ; dagg = Wn1a^T dnpre -> B0 (h is dead)
    node_gemm<HP, EPI_ACCUM, true>(wb, Lw.Wn1ht, B4, -1, nullptr, nullptr, dh, dh, nullptr, mg.NC, wave, lane, &pf, Lw.Wn1at);
    node_gemm<HP, EPI_NONE, true>(wb, Lw.Wn1at, B4, -1, nullptr, nullptr, B0, nullptr, nullptr, mg.NC, wave, lane, &pf);
    __syncthreads();
    for (int idx = tid; idx < N * LD; idx += kThreads) B4[idx] = 0.f;  // dQ accumulator
    __syncthreads();
    STAMP(ST_BWD_NODE);
    // (e) edge pass, all four waves in lock step.  Each wave runs the MLP chain forward and backward for NB
    //     16-edge tiles at a time (registers), then publishes the NB du tiles one after the other through its
    //     single scratch tile (two workgroup barriers per published tile).
    {
      constexpr int NB = GAUDI_BWD_TILES;
      SegSum<HP> ss;
      ss.init();
      float* scr = sm.scr + wave * 16 * LD;
      const int rounds = (ntmax + NB - 1) / NB;
      for (int rd = 0; rd < rounds; ++rd) {
        const int tile0 = rd * NB;
        const bool active = tile0 < nt_me;  // npairs*2 tiles: with NB <= 2 a round is all-active or all-idle
        f4 du[NB][T];
        int my_i[NB];
#pragma unroll
        for (int e = 0; e < NB; ++e) my_i[e] = 0;
        if (active) {
          EdgeCol ec[NB];
          float mk[NB];
          f4 gg[NB];
          load_cols<SM, NB>(sm, mg, wave, tile0 * 16, c, ec, mk, gg);
          // cpre, the attention gate and (after the first GEMM) v come back from the forward's stash: no recompute of W2 / Wc1
          // and only one NB x T quad array (the GEMM input) live in VGPRs across each GEMM.
          constexpr bool kKeepV = false;
          f4 v[kKeepV ? NB : 1][T], cp[NB][T];
          STAMP(ST_BWD_EDGE);
          float a[NB], tau[NB], dtx[NB], dty[NB], dtz[NB];
#pragma unroll
          for (int e = 0; e < NB; ++e) {
            const f4* sc = (const f4*)(estash + edge_stash_off(l, wave, tile0 + e, 1, EW, HP)) + lane;
            if (!last) {
#pragma unroll
              for (int t = 0; t < T; ++t) cp[e][t] = stash_load(sc + t * 64);
            }
            my_i[e] = ec[e].i;
            a[e] = astash[((size_t)l * kWaves + wave) * EW + (tile0 + e) * 16 + c];
            tau[e] = 0.f;
            dtx[e] = sm.dx[4 * ec[e].i + 0];  // dtrans = dx'_i
            dty[e] = sm.dx[4 * ec[e].i + 1];
            dtz[e] = sm.dx[4 * ec[e].i + 2];
          }
          STAMP(ST_B_V);
          STAMP(ST_B_EV);
          f4 de[NB][T];
          if (!last) {
            const float* rowinit[NB];
#pragma unroll
            for (int e = 0; e < NB; ++e) rowinit[e] = B0 + ec[e].i * LD;
            STAMP(ST_B_CP);
#pragma unroll
            for (int e = 0; e < NB; ++e) {
              float sd2 = 0.f;
#pragma unroll
              for (int t = 0; t < T; ++t) sd2 += dot4(silu4(cp[e][t]), *(const f4*)(Lw.wc2 + 16 * t + 4 * g));
              const float phi = reduce_groups(sd2);
              const float th = tanhf(phi);
              tau[e] = W.use_tanh ? th * W.coords_range_layer : phi;
              const float dtau = (dtx[e] * gg[e][1] + dty[e] * gg[e][2] + dtz[e] * gg[e][3]) * mk[e];
              const float dphi = W.use_tanh ? dtau * W.coords_range_layer * (1.0f - th * th) : dtau;
#pragma unroll
              for (int t = 0; t < T; ++t)  // dcpre = dphi * wc2 * silu'(cpre)
                cp[e][t] = *(const f4*)(Lw.wc2 + 16 * t + 4 * g) * dphi * dsilu4(cp[e][t]);
            }
            STAMP(ST_B_DCP);
            edge_gemm_from_regs<HP, NB>(de, cp, wb, Lw.Wc1t, nullptr, rowinit, lane);  // + dagg_i (agg_i = sum_j e_ij)
            STAMP(ST_B_DE);
          } else {
#pragma unroll
            for (int e = 0; e < NB; ++e)
#pragma unroll
              for (int t = 0; t < T; ++t) de[e][t] = *(const f4*)(B0 + ec[e].i * LD + 16 * t + 4 * g);
          }
#pragma unroll
          for (int e = 0; e < NB; ++e) {
            // e = m * a * mask ; a = sigmoid(wa . m + ba)
            f4(&ve)[T] = v[kKeepV ? e : 0];
            if (!kKeepV) {
              const f4* sv = (const f4*)(estash + edge_stash_off(l, wave, tile0 + e, 0, EW, HP)) + lane;
#pragma unroll
              for (int t = 0; t < T; ++t) ve[t] = stash_load(sv + t * 64);
            }
            float dadot = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) dadot += dot4(de[e][t], silu4(ve[t]));
            const float da = reduce_groups(dadot) * mk[e];
            const float ds = W.attention ? da * a[e] * (1.0f - a[e]) : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)  // dv = (de*a*mask + ds*wa) * silu'(v)
              de[e][t] = (de[e][t] * a[e] * mk[e] + *(const f4*)(Lw.wa + 16 * t + 4 * g) * ds) * dsilu4(ve[t]);
          }
          {
            const float* noinit[NB];
#pragma unroll
            for (int e = 0; e < NB; ++e) noinit[e] = nullptr;
            STAMP(ST_B_DV);
            edge_gemm_from_regs<HP, NB>(du, de, wb, Lw.W2t, nullptr, noinit, lane);  // dt1
            STAMP(ST_B_DT1);
          }
#pragma unroll
          for (int e = 0; e < NB; ++e) {
            // du = dt1 * silu'(u) ; dr = cr . du ; dd0 = cd . du
            const float* pp = B2 + ec[e].i * LD + 4 * g;
            const float* qq = B1 + ec[e].j * LD + 4 * g;
            float drdot = 0.f, dd0dot = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const f4 u = edge_u(pp, qq, Lw.cr, Lw.cd, g, t, ec[e].r, ec[e].d0);
              du[e][t] = du[e][t] * dsilu4(u);
              drdot += dot4(du[e][t], *(const f4*)(Lw.cr + 16 * t + 4 * g));
              dd0dot += dot4(du[e][t], *(const f4*)(Lw.cd + 16 * t + 4 * g));
            }
            const float dr = reduce_groups(drdot), dd0v = reduce_groups(dd0dot);
            if (g == 0) {
              // d/d(diff) of r = |diff|^2 and dhat = diff / (sqrt(r + 1e-8) + 1)   (gcl.py:308-316)
              const int i = ec[e].i, j = ec[e].j;
              const float fx = sm.x[4 * i + 0] - sm.x[4 * j + 0];
              const float fy = sm.x[4 * i + 1] - sm.x[4 * j + 1];
              const float fz = sm.x[4 * i + 2] - sm.x[4 * j + 2];
              const float nrm = sqrtf(gg[e][0] + 1e-8f), den = nrm + 1.0f;
              const float cx = dtx[e] * tau[e] * mk[e], cy = dty[e] * tau[e] * mk[e], cz = dtz[e] * tau[e] * mk[e];
              const float k1 = (cx * fx + cy * fy + cz * fz) / (den * den * nrm);
              const int slot = wave * EW + (tile0 + e) * 16 + c;
              sm.trans[4 * slot + 0] = cx / den - fx * k1 + 2.0f * fx * dr;
              sm.trans[4 * slot + 1] = cy / den - fy * k1 + 2.0f * fy * dr;
              sm.trans[4 * slot + 2] = cz / den - fz * k1 + 2.0f * fz * dr;
              sm.dd0[slot] += dd0v;
            }
          }
        }
        STAMP(ST_B_DU);
        // publish the NB du tiles one by one: dP_i = sum_j du_ij (own rows), dQ_j = sum_i du_ij (owner of j)
#pragma unroll
        for (int e = 0; e < NB; ++e) {
          const int tile = tile0 + e;
          if (active) {
#pragma unroll
            for (int t = 0; t < T; ++t) *(f4*)(scr + c * LD + 16 * t + 4 * g) = du[e][t];
          }
          wave_lds_fence();
          __syncthreads();  // every wave's du tile of this round is in its scratch
          STAMP(ST_BWD_BARRIER);
          if (active) ss.add_tile(scr, my_i[e], B2, 1.0f, lane);  // overwrites P_i: dead by now
          {
            // lane l holds the sending node j and its owner wave for slot (wave l>>4, column l&15) of this round;
            // the 64 slots are then walked with v_readlane (no LDS traffic on the serial path)
            const int w2l = lane >> 4;
            const bool live = tile < 2 * mg.npairs_all_lane(w2l);
            const uint32_t ew = live ? mg.edge[w2l * EW + tile * 16 + (lane & 15)] : 0u;
            const int jl = (int)((ew >> 8) & 255);
            const int ownl = live ? (int)(mg.seg[jl] >> 30) : -1;
#pragma unroll
            for (int sl = 0; sl < 64; ++sl) {
              const int owner = __builtin_amdgcn_readlane(ownl, sl);
              if (owner == wave) {
                const int jj = __builtin_amdgcn_readlane(jl, sl);
                if (lane < HP / 4) {
                  const float* scr2 = sm.scr + (sl >> 4) * 16 * LD + (sl & 15) * LD;
                  f4* dst = (f4*)(B4 + jj * LD + 4 * lane);
                  *dst = *dst + *(const f4*)(scr2 + 4 * lane);
                }
              }
            }
          }
          wave_lds_fence();
          STAMP(ST_BWD_COL);
          __syncthreads();  // scratch may be overwritten
          STAMP(ST_BWD_BARRIER);
        }
      }
      node_prefetch<HP>(pf, wb, Lw.At, wave, lane);  // (f)'s first tiles travel across the barriers below
      if (nt_me > 0) ss.flush(B2, 1.0f, lane);
    }
    __syncthreads();
    // nodes without live edges never had their P row replaced by dP = 0
    for (int idx = tid; idx < N * LD; idx += kThreads) {
      const int n = idx / LD;
      if ((mg.seg[n] & 0x7fff) == 0) B2[idx] = 0.f;
    }
    // dx <- dx*mask + sum_{e: i=n} ddiff_e - sum_{e: j=n} ddiff_e : one thread per (wave list, node, axis) walks that
    // wave's slots in order, the four partial sums are then added in wave order (fixed order, no atomics; a single
    // thread per (node, axis) walking all 4 lists cost 4 % of the guided step)
    float* part = sm.scr;  // [kWaves][N*3]; the transposition scratch is idle between edge passes
    for (int t = tid; t < kWaves * N * 3; t += kThreads) {
      const int w2 = t / (N * 3), nd = t % (N * 3), n = nd / 3, d = nd % 3;
      float acc = 0.f;
      const int ns = 32 * mg.npairs_all_lane(w2);
      for (int s = 0; s < ns; ++s) {
        const uint32_t e = mg.edge[w2 * EW + s];
        const float v = sm.trans[4 * (w2 * EW + s) + d];
        if ((int)(e & 255) == n) acc += v;
        if ((int)((e >> 8) & 255) == n) acc -= v;
      }
      part[t] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      float acc = sm.dx[4 * n + d];
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) acc += part[w2 * N * 3 + idx];
      sm.dx[4 * n + d] = acc;
    }
    __syncthreads();
    // (f) dh += A^T dP + Bm^T dQ
    STAMP(ST_MISC);
    vec_prefetch(vpf, wb, PredLayerW::vec_off(lay.layer(l > 0 ? l - 1 : 0), HP), PredLayerW::vec_count(HP), tid);
    node_gemm<HP, EPI_ACCUM, true>(wb, Lw.At, B2, Lw.Bmt, B4, nullptr, dh, dh, nullptr, mg.NC, wave, lane, &pf,
                                   l > 0 ? lay.layer(l - 1) + 13 * HP * HP : -1);
    __syncthreads();
    STAMP(ST_BWD_NODE);
  }

  // embedding backward (time column dropped), d0 backward, input masking
  {
    const float* ew = w + lay.emb_w();
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      float acc = sm.dx[4 * n + d];
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2)
        for (int s = 0; s < 32 * mg.npairs_all[w2]; ++s) {
          const uint32_t e = mg.edge[w2 * EW + s];
          const int i = e & 255, j = (e >> 8) & 255;
          const float v = 2.0f * (sm.x0[4 * i + d] - sm.x0[4 * j + d]) * sm.dd0[w2 * EW + s];
          if (i == n) acc += v;
          if (j == n) acc -= v;
        }
      sGrad[n * D + d] = acc * mg.mask[n];
    }
    const float* msk = mg.mask;
    small_dots(
        N * F, HP, tid, [=](int q, int f) { return ew[f * F1 + q % F]; },
        [=](int q, int f) { return dh[(q / F) * LD + f]; },
        [=](int q, float acc) { sGrad[(q / F) * D + 3 + q % F] = acc * msk[q / F]; });
  }
  __syncthreads();
  STAMP(ST_PRED_IO);
}

// unit-test entry: pred (and optionally grad into sGrad) for z in sZ
template <int HP, bool GN = false>
__device__ __forceinline__ void predictor_entry(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad, float* sTmp,
                                float* sMean, float t_val, const float* dpred, bool want_grad, float* pred_out,
                                float readout_div, float* stash, int tid STAMP_DECL, float* gnode = nullptr) {
  (void)sTmp; (void)sMean;
  PredSmem<HP, GN> sm;
  sm.carve(net, mg.N, mg.EW, gnode);
  pred_forward<HP>(W, mg, sm, sZ, t_val, stash, readout_div, tid STAMP_ARGS);
  if (tid < W.K) {
    if (pred_out) pred_out[tid] = sm.pred[tid];
    sm.pred[16 + tid] = dpred ? dpred[tid] : 0.f;
  }
  __syncthreads();
  if (want_grad) pred_backward<HP>(W, mg, sm, stash, sGrad, readout_div, tid STAMP_ARGS);
}

// guidance of one reverse step (en_diffusion.py:899-920): z_s <- z_s - sigma * P(clip(grad))
// phase 0: fused (target linear in pred: dT/dpred = target_w);  phase 1: predictor forward only, pred -> pred_out
// (the host evaluates an arbitrary target on it);  phase 2: reverse pass + update with dT/dpred = dpred_ext.
template <int HP, bool GN = false>
__device__ __forceinline__ void guidance_update(const PredDev& W, const MolGraph& mg, float* net, float* sZ, float* sGrad, float* sTmp,
                                float* sMean, float t_val, float sigma, const float* target_w, float scale,
                                float* pred_out, float readout_div, float* stash, int tid STAMP_DECL, int phase = 0,
                                const float* dpred_ext = nullptr, float* gnode = nullptr, const float* dz_ext = nullptr) {
  (void)sTmp;
  const int N = mg.N, D = mg.D;
  PredSmem<HP, GN> sm;
  sm.carve(net, N, mg.EW, gnode);
  if (phase != 2) pred_forward<HP>(W, mg, sm, sZ, t_val, stash, readout_div, tid STAMP_ARGS);
  if (tid < W.K) {
    if (pred_out && phase != 2) pred_out[tid] = sm.pred[tid];
    // energy = scale * sum_b T(pred_b)  ->  d(energy)/dpred = scale * dT/dpred
    sm.pred[16 + tid] = (phase == 2 ? dpred_ext[tid] : target_w[tid]) * scale;
  }
  __syncthreads();
  if (phase == 1) return;
  pred_backward<HP>(W, mg, sm, stash, sGrad, readout_div, tid STAMP_ARGS, phase == 2 ? sZ : nullptr);
  if (dz_ext != nullptr) {  // + scale * dT/dz of a target that depends on z outside the predictor (gaudi_sample_cbz), before the clip
    for (int e = tid; e < N * D; e += kThreads) sGrad[e] += dz_ext[e];
    __syncthreads();
  }
  // clip_coef = min(1, 10 / (||g||_2 + 1e-6)) over all N*(3+F) entries   (en_diffusion.py:905-909)
  if (tid < 64) {
    float s = 0.f;
    for (int e = tid; e < N * D; e += 64) s += sGrad[e] * sGrad[e];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (tid == 0) sMean[4] = fminf(10.0f / (sqrtf(s) + 1e-6f), 1.0f);
  }
  __syncthreads();
  const float coef = sMean[4];
  for (int e = tid; e < N * D; e += kThreads) sGrad[e] *= coef;
  __syncthreads();
  if (tid < 3) {  // masked mean of the x part of the gradient (en_diffusion.py:911-919)
    float s = 0.f, cnt = 0.f;
    for (int n = 0; n < N; ++n) { s += sGrad[n * D + tid]; cnt += mg.mask[n]; }
    sMean[tid] = s / fmaxf(cnt, 1.0f);
  }
  __syncthreads();
  for (int e = tid; e < N * D; e += kThreads) {
    const int n = e / D, d = e % D;
    float gv = sGrad[e];
    if (d < 3) gv = gv - sMean[d] * mg.mask[n];
    sZ[e] = sZ[e] - sigma * gv;
  }
  __syncthreads();
  STAMP(ST_GUIDE);
}

}  // namespace gaudi
