// edm_device.h -- EGNN_dynamics._forward (edm/egnn/models.py:83-152, edm/egnn/egnn_new.py) for one
// molecule held by one workgroup.  Input z and output eps_hat live in LDS.
#pragma once
#include "device_common.h"

namespace gaudi {

// Packed EDM weight buffer (floats).  HP = hidden padded to 16, PK = HP*HP (tile-packed matrix).
//   head : emb_w [HP][F1] | emb_b [HP] | out_w [F1][HP] | out_b [16]
//   block: S x GCL { A, Bm, W2, Wn1h, Wn1a, Wn2 (6 PK) | cr, cd, b1, b2, wa, bn1, bn2 (7 HP) | ba (16) }
//          EqU     { A, Bm, W2 (3 PK) | cr, cd, b1, b2, w3 (5 HP) | max|cr|, max|cd| (16) }
//          (the GCL's 16-float block holds ba, max|cr|, max|cd|: the column maxima bound the split edge GEMMs' scales, w8_split.h)
// A/Bm/cr/cd are the column blocks of Linear(2H+2 -> H): W1 = [A | Bm | cr | cd].
//   sin_embedding checkpoints (EF = 24 edge features instead of 2, egnn_new.py:269-273): cr, cd become the EF columns C_0 .. C_23 of
//   W1 = [A | Bm | C]; the vector blocks grow to (5 + EF) HP + 16 and (3 + EF) HP + 16 floats
struct EdmLayout {
  int HP, F1, L, S;
  int EF = 2;  // edge features of the first Linear of every edge / coordinate MLP
  __host__ __device__ int pk() const { return HP * HP; }
  __host__ __device__ int emb_w() const { return 0; }
  __host__ __device__ int emb_b() const { return align16(HP * F1); }
  __host__ __device__ int out_w() const { return emb_b() + HP; }
  __host__ __device__ int out_b() const { return out_w() + align16(F1 * HP); }
  __host__ __device__ int blocks() const { return out_b() + 16; }
  __host__ __device__ int gcl_size() const { return 6 * pk() + (5 + EF) * HP + 16; }
  __host__ __device__ int equ_size() const { return 3 * pk() + (3 + EF) * HP + 16; }
  __host__ __device__ int block_size() const { return S * gcl_size() + equ_size(); }
  __host__ __device__ int gcl(int l, int s) const { return blocks() + l * block_size() + s * gcl_size(); }
  __host__ __device__ int equ(int l) const { return blocks() + l * block_size() + S * gcl_size(); }
  __host__ __device__ int total() const { return blocks() + L * block_size(); }
};

struct EdmDev {
  const float* w;
  unsigned w_bytes;
  int F, L, S, attention, use_tanh;
  float coords_range, norm_constant, normf;
  int ktail;  // 8-wave kernels: the last K chunk of every matrix holds 4 valid inputs, packed as ONE k-step (w8_common.h)
  const float* ws;     // fp16-pair images of the edge-GEMM matrices (w8_split.h) and, in the same buffer, the fp16-pair
  unsigned ws_bytes;   // images of the node-GEMM matrices (w8_nodes_f16.h); nullptr without them
  float hinv;          // 2^-s: descale of the fp16-pair node images (one exponent per network)
};

// Per-molecule graph metadata prepared on the host (gaudi_hip.hip: build_meta) and staged in LDS.
struct MolGraph {
  int N, D, EW;           // nodes (padded), 3+F, per-wave edge-slot capacity (multiple of 32)
  int NC;                 // node columns that matter: 1 + last node that is live or touches a live edge (<= N)
  const float* mask;      // LDS [N]
  const uint32_t* edge;   // LDS [4][EW]  i | j<<8
  const float* em;        // LDS [4][EW]  edge_mask value (0 for padding slots)
  const uint32_t* seg;    // LDS [N]      wave<<30 | start<<15 | len  (edge run of node n)
  int npairs;             // 32-edge passes of THIS wave
  int npairs_all[kWaves]; // ... of every wave of the workgroup (lock-step loops of the reverse pass)
  // npairs_all[w] for a per-lane (non-uniform) w without dynamically indexing the array (which would spill it)
  __device__ __forceinline__ int npairs_all_lane(int w) const {
    return w == 0 ? npairs_all[0] : w == 1 ? npairs_all[1] : w == 2 ? npairs_all[2] : npairs_all[3];
  }
};

// LDS working set of one network evaluation.  GN ("global node buffers"): the [N][HP+4] node buffers -- what outgrows 160 KiB
// of LDS beyond ~22 graph nodes at the default widths -- live in a per-molecule global scratch instead (L2-resident; every
// cross-wave hand-off of them already sits behind a workgroup barrier, which orders global memory inside a workgroup too).
// The code is the same: the pointers carry their address space, hipcc emits global loads / stores for them.  Slower, but it
// lifts the graph-size limit of the library to what the reference accepts (sampling_edm.py:172-209 has no cap).
// EF = 24 (sin_embedding): the vector block holds the 24 feature columns, and every edge slot its 24 sinusoid features
// (feat: [4][EW][EF], the d0 half written once per evaluation, the r half once per block)
template <int HP, bool GN = false, int EF = 2>
struct NetSmem {
  static constexpr bool kGlobalNodes = GN;
  static constexpr int kEF = EF;
  float* feat = nullptr;   // [4][EW][EF] (EF > 2 only)
  float *h, *p, *q, *agg;  // [N][HP+4]
  float* scr;              // [4][16][HP+4]  per-wave transposition scratch
  float *x, *x0;           // [N][4]
  f4* geo;                 // [4][EW] (r, dhat)
  float* d0;               // [4][EW]
  float* trans;            // [4][EW][4]
  float* vec;              // [8*HP] the current layer's small vectors (cr, cd, b1, b2, wa/w3, bn1, bn2, ba)
  __host__ __device__ static int floats(int N, int EW) {
    return (GN ? 0 : 4 * N * (HP + 4)) + kWaves * 16 * (HP + 4) + 8 * N + kWaves * EW * 9 + (6 + EF) * HP +
           (EF > 2 ? kWaves * EW * EF : 0);
  }
  __device__ void carve(float* base, int N, int EW, float* gnode = nullptr) {
    constexpr int LD = HP + 4;
    float*& nb = GN ? gnode : base;
    h = nb; nb += N * LD;
    p = nb; nb += N * LD;
    q = nb; nb += N * LD;
    agg = nb; nb += N * LD;
    scr = base; base += kWaves * 16 * LD;
    x = base; base += 4 * N;
    x0 = base; base += 4 * N;
    geo = (f4*)base; base += kWaves * EW * 4;
    d0 = base; base += kWaves * EW;
    trans = base; base += kWaves * EW * 4;
    vec = base; base += (6 + EF) * HP;
    if (EF > 2) feat = base;
  }
};

// SinusoidsEmbeddingNew (egnn_new.py:378-391): x -> sqrt(x + 1e-8) * f_k, (sin | cos), f_k = 2 pi 4^k / 15 (k = 0..5) in the fp32
// values torch builds (oracle/gaudi_oracle.py: sin_frequencies; tests/golden/g22 holds the reference's tensor).  Accurate sinf / cosf:
// the highest frequency multiplies sqrt(r) by 429.
__device__ __forceinline__ void sin_features(float x, float* out /* [12] */) {
  constexpr float kFreq[6] = {4.1887903e-01f, 1.6755161e+00f, 6.7020645e+00f, 2.6808258e+01f, 1.0723303e+02f, 4.2893213e+02f};
  const float sx = sqrtf(x + 1e-8f);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float e = sx * kFreq[k];
    out[k] = sinf(e);
    out[6 + k] = cosf(e);
  }
}

__device__ __forceinline__ void edge_ij(uint32_t e, int& i, int& j) { i = e & 255; j = (e >> 8) & 255; }

// r = |x_i - x_j|^2, dhat = (x_i - x_j) / (sqrt(r + 1e-8) + norm_constant)   (egnn_new.py:394-400)
template <class SM>
__device__ __forceinline__ void compute_geo(const SM& sm, const MolGraph& mg, float norm_constant, int wave,
                                            int lane, bool write_d0) {
  for (int slot = lane; slot < mg.npairs * 32; slot += 64) {
    int i, j;
    edge_ij(mg.edge[wave * mg.EW + slot], i, j);
    const float dx = sm.x[4 * i + 0] - sm.x[4 * j + 0];
    const float dy = sm.x[4 * i + 1] - sm.x[4 * j + 1];
    const float dz = sm.x[4 * i + 2] - sm.x[4 * j + 2];
    const float r = dx * dx + dy * dy + dz * dz;
    if (write_d0) {
      sm.d0[wave * mg.EW + slot] = r;
    } else {
      const float inv = 1.0f / (sqrtf(r + 1e-8f) + norm_constant);
      sm.geo[wave * mg.EW + slot] = (f4){r, dx * inv, dy * inv, dz * inv};
    }
    if constexpr (SM::kEF > 2) {  // edge_attr = [sin_embedding(r) | sin_embedding(d0)] (egnn_new.py:217-219, 302-303)
      float* ft = sm.feat + (size_t)(wave * mg.EW + slot) * SM::kEF + (write_d0 ? SM::kEF / 2 : 0);
      sin_features(r, ft);
    }
  }
}

template <class SM, int NE>
__device__ __forceinline__ void load_cols(const SM& sm, const MolGraph& mg, int wave, int first_slot, int c,
                                          EdgeCol (&ec)[NE], float (&mk)[NE], f4 (&geo)[NE]) {
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int s = wave * mg.EW + first_slot + 16 * e + c;
    edge_ij(mg.edge[s], ec[e].i, ec[e].j);
    mk[e] = mg.em[s];
    geo[e] = sm.geo[s];
    ec[e].r = geo[e][0];
    ec[e].d0 = sm.d0[s];
    if constexpr (SM::kEF > 2) ec[e].ft = sm.feat + (size_t)s * SM::kEF;
  }
}

// x <- (x + sum_j trans_ij / normf) * mask     (egnn_new.py:132-155), fixed ascending-j order
template <class SM>
__device__ __forceinline__ void coord_update(const SM& sm, const MolGraph& mg, float normf, int tid) {
  for (int idx = tid; idx < mg.N * 3; idx += kThreads) {  // strided: N * 3 may exceed the workgroup at small hidden sizes
    const int n = idx / 3, d = idx % 3;
    const uint32_t sg = mg.seg[n];
    const int w = sg >> 30, st = (sg >> 15) & 0x7fff, len = sg & 0x7fff;
    float s = 0.f;
    for (int k = 0; k < len; ++k) s += sm.trans[(w * mg.EW + st + k) * 4 + d];
    sm.x[4 * n + d] = (sm.x[4 * n + d] + s / normf) * mg.mask[n];
  }
}

// eps_hat[N][D] (LDS) = EGNN_dynamics._forward(t, z[N][D] (LDS))
template <int HP, class SM = NetSmem<HP>>
__device__ __forceinline__ void edm_forward(const EdmDev& W, const MolGraph& mg, const SM& sm, const float* sZ, float* sEps,
                            float* sMean /* [4] */, float t_val, int tid STAMP_DECL) {
  constexpr int LD = HP + 4;
  constexpr int T = HP / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int N = mg.N, D = mg.D, F = W.F, F1 = F + 1;
  constexpr int EF = SM::kEF;  // edge features: 2, or 24 for sin_embedding checkpoints
  EdmLayout lay{HP, F1, W.L, W.S, EF};
  const float* __restrict__ w = W.w;
  const WBuf wb = make_wbuf(W.w, W.w_bytes);

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
      for (int k = 0; k < F; ++k) acc += ew[f * F1 + k] * (sZ[n * D + 3 + k] * m);
      acc += ew[f * F1 + F] * t_val;
      sm.h[n * LD + f] = acc + eb[f];
    }
  }
  __syncthreads();
  compute_geo(sm, mg, 0.f, wave, lane, true);  // d0 of the input coordinates (egnn_new.py:301)
  NodePF<HP> pf;  // first weight tiles of the next node GEMM, loaded ahead of it (device_common.h)
  node_prefetch<HP>(pf, wb, lay.gcl(0, 0), wave, lane);
  constexpr int kVG = (5 + EF) * HP + 16, kVE = (3 + EF) * HP;  // floats of a GCL's / an EquivariantUpdate's vector block
  constexpr int NV = (kVG + kThreads - 1) / kThreads;
  VecPF<NV> vpf;  // the next sub-layer's vectors, loaded a phase ahead
  vec_prefetch(vpf, wb, lay.gcl(0, 0) + 6 * HP * HP, kVG, tid);
  STAMP(ST_EDM_IO);

  for (int l = 0; l < W.L; ++l) {
    compute_geo(sm, mg, W.norm_constant, wave, lane, false);  // egnn_new.py:216
    for (int s = 0; s < W.S; ++s) {
      // ------------------------------------------------------------------ GCL (egnn_new.py:42-89)
      const int G = lay.gcl(l, s);  // float offsets into the weight buffer
      const int PK = HP * HP;
      const int V = G + 6 * PK;
      // stage the layer's vectors in LDS (previous readers are behind the barrier that ended the last layer)
      (void)V;
      vec_commit(vpf, sm.vec, kVG, tid);
      __syncthreads();
      STAMP(ST_STAGE);
      const float *cr = sm.vec, *cd = sm.vec + HP, *b1 = sm.vec + EF * HP, *b2 = sm.vec + (EF + 1) * HP, *wa = sm.vec + (EF + 2) * HP,
                  *bn1 = sm.vec + (EF + 3) * HP, *bn2 = sm.vec + (EF + 4) * HP;
      const float ba = sm.vec[(EF + 5) * HP];
      node_gemm<HP, EPI_NONE, true>(wb, G, sm.h, -1, nullptr, b1, sm.p, nullptr, nullptr, mg.NC, wave, lane, &pf, G + PK);
      node_gemm<HP, EPI_NONE, true>(wb, G + PK, sm.h, -1, nullptr, nullptr, sm.q, nullptr, nullptr, mg.NC, wave, lane, &pf);
      for (int idx = tid; idx < N * LD; idx += kThreads) sm.agg[idx] = 0.f;
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
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
          edge_gemm_from_pq<HP, 2, EF>(acc, wb, G + 2 * PK, b2, cr, cd, sm.p, sm.q, ec, lane);
          STAMP(ST_EDGE);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const f4 m = silu4(acc[e][t]);
              acc[e][t] = m;
              const f4 wv = *(const f4*)(wa + 16 * t + 4 * g);
              sdot += m[0] * wv[0] + m[1] * wv[1] + m[2] * wv[2] + m[3] * wv[3];
            }
            float a = 1.f;
            if (W.attention) a = sigmoid_f(reduce_groups(sdot) + ba);
            const float mk = mk2[e];
#pragma unroll
            for (int t = 0; t < T; ++t) *(f4*)(scr + c * LD + 16 * t + 4 * g) = acc[e][t] * a * mk;
            wave_lds_fence();
            ss.add_tile(scr, ec[e].i, sm.agg, W.normf, lane);
            wave_lds_fence();
          }
        }
        node_prefetch<HP>(pf, wb, G + 3 * PK, wave, lane);  // node MLP weights travel across the barrier
        ss.flush(sm.agg, W.normf, lane);
      }
      STAMP(ST_EDGE_EPI);
      __syncthreads();
      STAMP(ST_BARRIER);
      node_gemm<HP, EPI_SILU, true>(wb, G + 3 * PK, sm.h, G + 4 * PK, sm.agg, bn1, sm.p, nullptr, nullptr, mg.NC, wave, lane,
                                    &pf, G + 5 * PK);
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
      vec_prefetch(vpf, wb, s + 1 < W.S ? lay.gcl(l, s + 1) + 6 * PK : lay.equ(l) + 3 * PK, s + 1 < W.S ? kVG : kVE, tid);
      node_gemm<HP, EPI_RESIDUAL_MASK, true>(wb, G + 5 * PK, sm.p, -1, nullptr, bn2, sm.h, sm.h, mg.mask, mg.NC, wave, lane,
                                             &pf, s + 1 < W.S ? lay.gcl(l, s + 1) : lay.equ(l));
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
    }
    // -------------------------------------------------------- EquivariantUpdate (egnn_new.py:119-155)
    {
      const int E = lay.equ(l);
      const int PK = HP * HP;
      const int V = E + 3 * PK;
      (void)V;
      vec_commit(vpf, sm.vec, kVE, tid);
      __syncthreads();
      STAMP(ST_STAGE);
      const float *cr = sm.vec, *cd = sm.vec + HP, *b1 = sm.vec + EF * HP, *b2 = sm.vec + (EF + 1) * HP, *w3 = sm.vec + (EF + 2) * HP;
      node_gemm<HP, EPI_NONE, true>(wb, E, sm.h, -1, nullptr, b1, sm.p, nullptr, nullptr, mg.NC, wave, lane, &pf, E + PK);
      node_gemm<HP, EPI_NONE, true>(wb, E + PK, sm.h, -1, nullptr, nullptr, sm.q, nullptr, nullptr, mg.NC, wave, lane, &pf);
      STAMP(ST_NODE);
      __syncthreads();
      STAMP(ST_BARRIER);
      for (int tp = 0; tp < mg.npairs; ++tp) {
        EdgeCol ec[2];
        float mk2[2];
        f4 geo2[2];
        load_cols<SM, 2>(sm, mg, wave, tp * 32, c, ec, mk2, geo2);
        f4 acc[2][T];
        edge_gemm_from_pq<HP, 2, EF>(acc, wb, E + 2 * PK, b2, cr, cd, sm.p, sm.q, ec, lane);
        STAMP(ST_EDGE);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float sdot = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f4 m = silu4(acc[e][t]);
            const f4 wv = *(const f4*)(w3 + 16 * t + 4 * g);
            sdot += m[0] * wv[0] + m[1] * wv[1] + m[2] * wv[2] + m[3] * wv[3];
          }
          const float phi = reduce_groups(sdot);
          const float tau = W.use_tanh ? tanhf(phi) * W.coords_range : phi;
          const f4 gg = geo2[e];
          const float mk = mk2[e];
          if (g == 0) {
            const int slot = wave * mg.EW + tp * 32 + e * 16 + c;
            *(f4*)(sm.trans + 4 * slot) = (f4){gg[1] * tau * mk, gg[2] * tau * mk, gg[3] * tau * mk, 0.f};
          }
        }
      }
      if (l + 1 < W.L) {
        node_prefetch<HP>(pf, wb, lay.gcl(l + 1, 0), wave, lane);
        vec_prefetch(vpf, wb, lay.gcl(l + 1, 0) + 6 * HP * HP, kVG, tid);  // travels across the barrier and coord_update
      }
      STAMP(ST_EDGE_EPI);
      __syncthreads();
      STAMP(ST_BARRIER);
      coord_update(sm, mg, W.normf, tid);
      __syncthreads();
      STAMP(ST_MISC);
    }
  }

  // ---- head: embedding_out * mask (egnn_new.py:316-318); vel = (x - x_in) * mask, NaN -> 0,
  //      masked mean removal (models.py:116-152); the time column of h is dropped.
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
    small_dots(
        N * F, HP, tid, [=](int p, int k) { return ow[(p % F) * HP + k]; },
        [=](int p, int k) { return hh[(p / F) * LD + k]; },
        [=](int p, float acc) { sEps[(p / F) * D + 3 + p % F] = (acc + ob[p % F]) * msk[p / F]; });
    // `if torch.any(torch.isnan(vel)): vel = torch.nan_to_num(vel, 0.0)` (models.py:138-141): NaN -> 0, +-inf -> +-FLT_MAX,
    // triggered per molecule here (the reference looks at the whole batch; see sampler_kernel.h for the one difference)
    if (__syncthreads_or(bad)) {
      for (int idx = tid; idx < N * 3; idx += kThreads) {
        const int n = idx / 3, d = idx % 3;
        const float v = sEps[n * D + d];
        sEps[n * D + d] = v != v ? 0.f : fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
      }
      __syncthreads();
    }
    if (tid < 3) {
      float s = 0.f, cnt = 0.f;
      for (int n = 0; n < N; ++n) {
        s += sEps[n * D + tid];
        cnt += mg.mask[n];
      }
      sMean[tid] = s / fmaxf(cnt, 1.0f);
    }
    __syncthreads();
    for (int idx = tid; idx < N * 3; idx += kThreads) {
      const int n = idx / 3, d = idx % 3;
      sEps[n * D + d] = sEps[n * D + d] - sMean[d] * mg.mask[n];
    }
    __syncthreads();
  }
  STAMP(ST_EDM_IO);
}

}  // namespace gaudi
