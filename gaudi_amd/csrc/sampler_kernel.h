// sampler_kernel.h -- the per-molecule persistent workgroup kernel: runs reverse-diffusion steps
// s_hi..s_lo (EDM denoiser + z update [+ predictor forward/backward + guidance]) for one molecule
// without leaving the CU, plus the unit-test modes (phi / predictor / decode).
#pragma once
#include "edm_device.h"
#include "pred_device.h"
#include "w8_edm.h"
#include "w8_pred.h"

#ifndef GAUDI_STATIC_PRIO
#define GAUDI_STATIC_PRIO 0
#endif

namespace gaudi {

enum Mode { MODE_PHI = 0, MODE_SAMPLE = 1, MODE_PRED_FWD = 2, MODE_PRED_GRAD = 3, MODE_GUIDE = 4 };

struct KParams {
  int mode, B, N, F, EW;   // B workgroups (molecules, or groups of them), N node slots per workgroup
  int NR;                   // rows per molecule in the global [molecules][NR][.] arrays (= N unless the launch runs wide groups:
                            // gaudi_hip.hip, stage_graph8 -- a group of several molecules then has MORE node slots than a molecule)
  int do_init, do_decode, guided;
  int s_hi, s_lo, T;
  // graph metadata (device)
  const float* node_mask;   // [B][N]
  const int* order;         // [B] block -> molecule (heaviest first)
  const uint32_t* edges;    // [B][4][EW]
  const float* emask;       // [B][4][EW]
  const int* npairs;        // [B][4]
  const uint32_t* seginfo;  // [B][N]
  const int* ncols;         // [B] node columns the node-level GEMMs have to produce (<= N)
  // 8-wave kernels (w8_*.h): edges / emask are flat [B][EW] (EW = slot capacity S), npairs = [B] 16-slot tiles per
  // molecule, seginfo = start << 16 | len, plus the sender lists of the reverse pass
  const uint16_t* soff;     // [B][N+1]
  const uint16_t* sidx;     // [B][EW]
  int pubx, pub_ch;         // reverse pass: extra floats of the du publish buffer, 16-feature tiles published per chunk
  // tensors (device)
  const float* z_in;        // [B][N][D]
  float* z_out;             // [B][N][D]
  const float* t_in;        // [B]
  float* x_out;             // [B][N][3]
  float* h_out;             // [B][N][F]
  const float* noise;       // injected raw draws or nullptr
  int draw_base;
  long long draw_stride;
  unsigned long long seed;
  long long sample_offset;
  int fix_noise;            // en_diffusion.py:562-566: every molecule takes the raw draws of ONE sample ...
  long long fix_key;        // ... the Philox stream of this global sample index (or row 0 of the injected buffer)
  float std0;
  const float* coef;        // [T][4] alpha_ts, eps_coef, sigma, t
  float alpha0, sigma0, sigma_x, nv0, nv1;
  int* nan_count;
  EdmDev edm;
  PredDev pred;
  const float* dpred_in;    // [B][K]  (MODE_PRED_GRAD) or nullptr
  const float* dz_in;       // [B][N][D] or nullptr: phase B of a callback step (split = 2) adds this to the reverse pass's gradient
                            // before the clip -- scale * dT/dz of a target that depends on z outside the predictor (gaudi_sample_cbz)
  const float* target_w;    // [K]     (guided sampling)
  float scale;
  float* pred_out;          // [B][K]
  float* stash;             // predictor activation stash, [B] x stash_stride floats
  long long stash_stride;
  float readout_div;        // padded N the predictor readout divides by
  const float* alpha_sigma; // [B][2] or nullptr: forward-noising prologue z_t = alpha*normalize(xh) + sigma*eps (sample_edm_t)
  float* zt_out;            // [B][N][D] noised input (only with alpha_sigma) or nullptr
  int split;                // 0 fused step; 1 = phase A (denoise + predictor forward); 2 = phase B (reverse pass + update)
  float* chain_out;         // sample_chain: [keep_frames][B][N][D] un-normalised frames, or nullptr
  int keep_frames;
  unsigned long long* stamps;  // diagnostic builds only (-DGAUDI_STAMPS): [ST_N] cycle sums of block 0
  float* gnode;                // V4G kernels: [B] x gnode_stride floats, the node buffers of large molecules (edm_device.h: NetSmem GN)
  long long gnode_stride;
  // Packed launches (gaudi_hip.hip: pack_groups): a workgroup holds up to kMaxComp small molecules as ONE graph with
  // disconnected components; B = groups, every per-graph array above (node_mask, edges, ...) is per group.
  const int32_t* rowmap;       // [B][N] node slot -> global row (molecule * N + its node) | component << 28, or -1; nullptr = not packed
  const int32_t* compmol;      // [B][kMaxComp] molecule index of each component
  const int32_t* ncomp;        // [B]
  // Profiled launches (gaudi_profile_reset(h, 1)): lane 0 of workgroup 0 leaves the shader-clock and the constant 100 MHz
  // counters at kernel entry [0], [1] and at the end of its last step [2], [3] -- the clock the chip HELD under this kernel's
  // load (1.7-2.0 GHz against the nominal 2.4 the roofline peaks assume), bench.py: roofline.clock_mhz.  nullptr otherwise.
  unsigned long long* clock_out;
  int hk_off;                  // 8-wave split kernels: float offset (from the start of LDS) of the kept split copy of h, 0 = none
                               // (w8_nodes_f16.h: node_ctx_keep; placed by the host behind the whole plan when 160 KiB leave the room)
};
constexpr int kMaxComp = 4;

__host__ __device__ inline int common_floats(int N, int D, int EW) {
  return 3 * align16(N * D) + align16(N) + 16 + align16(N) + 16 + 2 * kWaves * EW + align16(N);
}
__host__ __device__ inline int common_floats8(int N, int D, int S) {
  return 3 * align16(N * D) + align16(N) + 16 + align16(N) + 16 + 2 * S + align16(N) + align16((N + 1 + S + 1) / 2);
}
// node slot -> row word (sRow): global row in bits 0-27, component in bits 28-30; -1 = empty slot (component 7)
__device__ __forceinline__ int row_of(int w) { return w & 0x0fffffff; }
__device__ __forceinline__ int comp_of(int w) { return (w >> 28) & 7; }

// ---- kernel variants: what differs between the 4-wave kernels (one wave per SIMD, per-wave edge lists, weights streamed
// per wave) and the 8-wave kernels (two waves per SIMD, flat 16-slot tiles, LDS-shared weight ring) behind one sampler body
// EF: edge features of the denoiser's first Linears -- 24 for sin_embedding checkpoints (edm_device.h), which only this family runs
template <bool GN, int EF = 2>
struct V4T {
  static constexpr int kThreads = gaudi::kThreads;
  static constexpr bool kGlobalNodes = GN;
  using Graph = gaudi::MolGraph;
  template <int HP> using EdmSmem = gaudi::NetSmem<HP, GN, EF>;
  __device__ __forceinline__ static void set_rows(Graph&, const int*, int, int) {}  // the 4-wave kernels are never packed
  __host__ __device__ static int graph_floats(int N, int EW) { return 2 * gaudi::kWaves * EW + align16(N); }
  __device__ __forceinline__ static float* load_graph(const KParams& P, int b, float* base, const float* sMask, Graph& mg, int tid, int wave) {
    const int N = P.N, EW = P.EW;
    uint32_t* sEdge = (uint32_t*)base; base += gaudi::kWaves * EW;
    float* sEm = base; base += gaudi::kWaves * EW;
    uint32_t* sSeg = (uint32_t*)base; base += align16(N);
    for (int i = tid; i < N; i += kThreads) sSeg[i] = P.seginfo[b * N + i];
    for (int i = tid; i < gaudi::kWaves * EW; i += kThreads) {
      sEdge[i] = P.edges[(size_t)b * gaudi::kWaves * EW + i];
      sEm[i] = P.emask[(size_t)b * gaudi::kWaves * EW + i];
    }
    mg.N = N; mg.D = 3 + P.F; mg.EW = EW;
    mg.NC = P.ncols[b];
    mg.mask = sMask; mg.edge = sEdge; mg.em = sEm; mg.seg = sSeg;
    mg.npairs = P.npairs[b * gaudi::kWaves + wave];
#pragma unroll
    for (int w = 0; w < gaudi::kWaves; ++w) mg.npairs_all[w] = P.npairs[b * gaudi::kWaves + w];
    return base;
  }
  // gnode: this workgroup's slice of the global node-buffer scratch (V4G), computed by the caller from the kernel arguments
  // right at the call (kept out of MolGraph: one more long-lived pointer made the largest instantiation fault)
  template <int HP>
  __device__ __forceinline__ static void edm(const EdmDev& W, const Graph& mg, float* net, const float* sZ, float* sEps,
                                             float* sMean, float t_val, int tid STAMP_DECL, float* gnode) {
    gaudi::NetSmem<HP, GN, EF> sm;
    sm.carve(net, mg.N, mg.EW, gnode);
    gaudi::edm_forward<HP>(W, mg, sm, sZ, sEps, sMean, t_val, tid STAMP_ARGS);
  }
  // explicit by-value signatures: forwarding references (and a by-reference KParams) made hipcc keep the arguments in
  // scratch in the largest instantiations, with wrong results / null dereferences on the GPU
  template <int HP>
  __device__ __forceinline__ static void guide(const PredDev& W, const Graph& mg, float* net, float* sZ, float* sGrad, float* sTmp,
                                               float* sMean, float t_val, float sigma, const float* target_w, float scale,
                                               float* pred_out, float readout_div, float* stash, int tid STAMP_DECL, int phase,
                                               const float* dpred_ext, float* gnode, const float* dz_ext) {
    gaudi::guidance_update<HP, GN>(W, mg, net, sZ, sGrad, sTmp, sMean, t_val, sigma, target_w, scale, pred_out, readout_div, stash,
                                   tid STAMP_ARGS, phase, dpred_ext, gnode, dz_ext);
  }
  template <int HP>
  __device__ __forceinline__ static void pred_entry(const PredDev& W, const Graph& mg, float* net, float* sZ, float* sGrad,
                                                    float* sTmp, float* sMean, float t_val, const float* dpred, bool want_grad,
                                                    float* pred_out, float readout_div, float* stash, int tid STAMP_DECL, float* gnode) {
    gaudi::predictor_entry<HP, GN>(W, mg, net, sZ, sGrad, sTmp, sMean, t_val, dpred, want_grad, pred_out, readout_div, stash,
                                   tid STAMP_ARGS, gnode);
  }
};
using V4 = V4T<false>;
using V4G = V4T<true>;  // node buffers in global memory: molecules beyond the LDS limit (N up to 255)
using V4S = V4T<false, 24>;  // sin_embedding denoiser (kern_se_*.hip)
using V4GS = V4T<true, 24>;

// ---- out-of-line phases of the 8-wave kernels.  Inlined into one 40k-instruction function, the denoiser, the predictor and
// its reverse pass are register-allocated together and hipcc spills ~200 VGPRs whose reloads land next to the deep weight
// prefetch queues (a scratch load shares the in-order vmcnt counter with them).  As separate functions each phase gets its
// own allocation; the call passes sizes and global pointers only and the callee re-derives every LDS pointer from the
// dynamic-LDS symbol itself, so LDS accesses keep their address space (ds_* instructions, not flat_*).
struct Lds8 {
  float *sZ, *sEps, *sNz, *sMask, *sMean, *net;
  int *sRow, *sCmol;  // node slot -> row word; [0..3] molecule of component k, [4] = number of components
  uint32_t *sEdge, *sSeg;
  float* sEm;
  uint16_t *sOff, *sIdx;
};
__device__ __forceinline__ Lds8 carve_lds8(float* smem, int N, int D, int S) {
  Lds8 L;
  float* base = smem;
  L.sZ = base; base += align16(N * D);
  L.sEps = base; base += align16(N * D);
  L.sNz = base; base += align16(N * D);
  L.sMask = base; base += align16(N);
  L.sMean = base; base += 16;
  L.sRow = (int*)base; base += align16(N);
  L.sCmol = (int*)base; base += 16;
  L.sEdge = (uint32_t*)base; base += S;
  L.sEm = base; base += S;
  L.sSeg = (uint32_t*)base; base += align16(N);
  L.sOff = (uint16_t*)base;
  L.sIdx = L.sOff + (N + 1);
  base += align16((N + 1 + S + 1) / 2);
  L.net = base;
  return L;
}
struct Graph8Args {
  int N, D, S, NC, ntiles, pubx, pub_ch, hk;
};
__device__ __forceinline__ w8::MolGraph graph8(const Lds8& L, const Graph8Args& a) {
  w8::MolGraph mg;
  mg.N = a.N; mg.D = a.D; mg.S = a.S; mg.NC = a.NC;
  mg.ntiles = a.ntiles;
  mg.rounds = (a.ntiles + w8::kWaves - 1) / w8::kWaves;
  mg.pubx = a.pubx; mg.pub_ch = a.pub_ch;
  mg.hk = a.hk;
  mg.mask = L.sMask; mg.edge = L.sEdge; mg.em = L.sEm; mg.seg = L.sSeg; mg.soff = L.sOff; mg.sidx = L.sIdx;
  mg.row = L.sRow;
  mg.ncomp = __builtin_amdgcn_readfirstlane(L.sCmol[4]);
  mg.NR = a.N;  // (the out-of-line phases never map a slot to its global row)
  return mg;
}
// Function arguments arrive in VGPRs: without these the callee treats every size, offset and buffer descriptor as
// divergent (vector ALU address arithmetic, a waterfall loop around every buffer_load).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uni(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <class T>
__device__ __forceinline__ T* uni(T* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ Graph8Args uni(const Graph8Args& a) {
  return Graph8Args{uni(a.N), uni(a.D), uni(a.S), uni(a.NC), uni(a.ntiles), uni(a.pubx), uni(a.pub_ch), uni(a.hk)};
}
__device__ __forceinline__ EdmDev uni(const EdmDev& w) {
  return EdmDev{uni(w.w), uni(w.w_bytes), uni(w.F), uni(w.L), uni(w.S), uni(w.attention), uni(w.use_tanh), uni(w.coords_range),
                uni(w.norm_constant), uni(w.normf), uni(w.ktail), uni(w.ws), uni(w.ws_bytes), uni(w.hinv)};
}
__device__ __forceinline__ PredDev uni(const PredDev& w) {
  return PredDev{uni(w.w), uni(w.w_bytes), uni(w.F), uni(w.K), uni(w.L), uni(w.attention), uni(w.use_tanh), uni(w.coords_range_layer), uni(w.ktail), uni(w.ws),
                 uni(w.ws_bytes), uni(w.hinv)};
}
#ifndef GAUDI_STAMPS
// GN: the node buffers of the phase live in the workgroup's slice of the global scratch (gnode_), everything else in LDS
template <int HP, int SP, int GN = 0, bool FL = false>
__device__ __attribute__((noinline)) void edm8_call(EdmDev W_, Graph8Args ga_, float t_val_, float* gnode_ = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const EdmDev W = uni(W_);
  const Graph8Args ga = uni(ga_);
  const float t_val = uni(t_val_);
  const Lds8 L = carve_lds8(smem, ga.N, ga.D, ga.S);
  const w8::MolGraph mg = graph8(L, ga);
  w8::NetSmem<HP, SP, GN> sm;
  sm.carve(L.net, ga.N, ga.S, GN ? uni(gnode_) : nullptr);
  sm.hk = ga.hk ? smem + ga.hk : nullptr;
  w8::edm_forward<HP, SP, GN, FL>(W, mg, sm, L.sZ, L.sEps, L.sMean, t_val, (int)threadIdx.x);
}
// the predictor's forward and reverse passes are separate functions too (the reverse pass holds three 52-register
// operand sets at its peak; allocated together with the forward it spilled twice as much)
template <int HP, int SP, bool MR, int GN = 0, bool FL = false, bool PG = false>
__device__ __attribute__((noinline)) void pred_fwd8_call(PredDev W_, Graph8Args ga_, float t_val_, float* stash_, float readout_div_,
                                                         float* gnode_ = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PredDev W = uni(W_);
  const Graph8Args ga = uni(ga_);
  const Lds8 L = carve_lds8(smem, ga.N, ga.D, ga.S);
  const w8::MolGraph mg = graph8(L, ga);
  w8::PredSmem<HP, SP, GN, PG> sm;
  sm.carve(L.net, ga.N, ga.S, ga.pubx, (GN || PG) ? uni(gnode_) : nullptr);
  sm.hk = ga.hk ? smem + ga.hk : nullptr;
  w8::pred_forward<HP, SP, MR, GN, FL>(W, mg, sm, L.sZ, uni(t_val_), uni(stash_), uni(readout_div_), (int)threadIdx.x);
}
template <int HP, int SP, bool MR, int GN = 0, bool FL = false, bool PG = false>
__device__ __attribute__((noinline)) void pred_bwd8_call(PredDev W_, Graph8Args ga_, float* stash_, float readout_div_, int resume_,
                                                         float* gnode_ = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PredDev W = uni(W_);
  const Graph8Args ga = uni(ga_);
  const Lds8 L = carve_lds8(smem, ga.N, ga.D, ga.S);
  const w8::MolGraph mg = graph8(L, ga);
  w8::PredSmem<HP, SP, GN, PG> sm;
  sm.carve(L.net, ga.N, ga.S, ga.pubx, (GN || PG) ? uni(gnode_) : nullptr);
  w8::pred_backward<HP, SP, MR, GN, FL>(W, mg, sm, uni(stash_), L.sEps /* grad */, uni(readout_div_), ga.pub_ch, (int)threadIdx.x,
                        uni(resume_) ? L.sZ : nullptr);
}
#endif

// SP: edge and node GEMMs on the fp16 matrix pipe with operands split into fp16 pairs (w8_split.h, w8_nodes_f16.h; three bf16 pieces
//     in rounds 2-4); otherwise fp32 MFMAs
// MR: the predictor takes graphs of more than one round of eight edge tiles (w8_pred.h); the denoiser always does
// GN: node buffers in the workgroup's global scratch (V8G, round 4: molecules beyond the LDS limit on the 8-wave kernels) -- 1: all
//     five; 2: P and Q stay in LDS (round 6, w8_edm.h: gn_lds_buffers)
// FR: the node GEMMs' split passes and epilogues recompute their lane addresses per call (w8_nodes_f16.h: FL) -- always in the MR
//     and GN kernels; the resident single-round kernel exists in both forms and the host picks by node slots (gaudi_hip.hip)
// PG: the predictor keeps ONE node buffer in the workgroup's global scratch (w8_pred.h: PredSmem) -- wide groups on the full ring
template <int SP, bool MR = false, int GN = 0, bool FR = false, bool PG = false>
struct V8T {
  static constexpr bool kFL = MR || GN != 0 || FR;
  static constexpr int kThreads = w8::kThreads;
  static constexpr int kSplit = SP;
  using Graph = w8::MolGraph;
  __host__ __device__ static int graph_floats(int N, int S) { return 2 * S + align16(N) + align16((N + 1 + S + 1) / 2); }
  __device__ __forceinline__ static float* load_graph(const KParams& P, int b, float* base, const float* sMask, Graph& mg, int tid, int wave) {
    (void)wave;
    const int N = P.N, S = P.EW;
    uint32_t* sEdge = (uint32_t*)base; base += S;
    float* sEm = base; base += S;
    uint32_t* sSeg = (uint32_t*)base; base += align16(N);
    uint16_t* sOff = (uint16_t*)base;
    uint16_t* sIdx = sOff + (N + 1);
    base += align16((N + 1 + S + 1) / 2);
    for (int i = tid; i < N; i += kThreads) sSeg[i] = P.seginfo[b * N + i];
    for (int i = tid; i < N + 1; i += kThreads) sOff[i] = P.soff[(size_t)b * (N + 1) + i];
    for (int i = tid; i < S; i += kThreads) {
      sEdge[i] = P.edges[(size_t)b * S + i];
      sEm[i] = P.emask[(size_t)b * S + i];
      sIdx[i] = P.sidx[(size_t)b * S + i];
    }
    mg.N = N; mg.D = 3 + P.F; mg.S = S;
    mg.NC = P.ncols[b];
    mg.ntiles = P.npairs[b];
    mg.rounds = (mg.ntiles + w8::kWaves - 1) / w8::kWaves;
    mg.pubx = P.pubx;
    mg.pub_ch = P.pub_ch;
    mg.hk = P.hk_off;
    mg.mask = sMask; mg.edge = sEdge; mg.em = sEm; mg.seg = sSeg; mg.soff = sOff; mg.sidx = sIdx;
    return base;
  }
  __device__ __forceinline__ static void set_rows(Graph& mg, const int* row, int ncomp, int NR) { mg.row = row; mg.ncomp = ncomp; mg.NR = NR; }
  __device__ __forceinline__ static Graph8Args gargs(const Graph& mg) {
    return Graph8Args{mg.N, mg.D, mg.S, mg.NC, mg.ntiles, mg.pubx, mg.pub_ch, mg.hk};
  }
  static constexpr bool kGlobalNodes = GN != 0 || PG;  // (the kernel passes its slice of the global scratch)
  template <int HP>
  __device__ __forceinline__ static void edm(const EdmDev& W, const Graph& mg, float* net, const float* sZ, float* sEps,
                                             float* sMean, float t_val, int tid STAMP_DECL, float* gnode) {
#ifdef GAUDI_STAMPS
    w8::NetSmem<HP, SP, GN> sm;
    sm.carve(net, mg.N, mg.S, GN ? uni(gnode) : nullptr);
    sm.hk = w8::lds_at(mg.hk);
    w8::edm_forward<HP, SP, GN>(W, mg, sm, sZ, sEps, sMean, t_val, tid STAMP_ARGS);
#else
    (void)net; (void)sZ; (void)sEps; (void)sMean; (void)tid;
    edm8_call<HP, SP, GN, kFL>(W, gargs(mg), t_val, gnode);
#endif
  }
  template <int HP>
  __device__ __forceinline__ static void guide(const PredDev& W, const Graph& mg, float* net, float* sZ, float* sGrad, float* sTmp,
                                               float* sMean, float t_val, float sigma, const float* target_w, float scale,
                                               float* pred_out, float readout_div, float* stash, int tid STAMP_DECL, int phase,
                                               const float* dpred_ext, float* gnode, const float* dz_ext) {
#ifdef GAUDI_STAMPS
    (void)dz_ext;  // (the stamped diagnostic build times the fused step only)
    w8::guidance_update<HP, SP, MR, GN, PG>(W, mg, net, sZ, sGrad, sTmp, sMean, t_val, sigma, target_w, scale, pred_out, readout_div, stash,
                                            mg.pubx, mg.pub_ch, tid STAMP_ARGS, phase, dpred_ext, (GN || PG) ? uni(gnode) : nullptr);
#else
    (void)sTmp;
    w8::PredSmem<HP, SP, GN, PG> sm;
    sm.carve(net, mg.N, mg.S, mg.pubx, gnode);
    if (phase != 2) pred_fwd8_call<HP, SP, MR, GN, kFL, PG>(W, gargs(mg), t_val, stash, readout_div, gnode);
    w8::guidance_seed(W, sm, target_w, scale, pred_out, tid, phase, dpred_ext);
    if (phase == 1) return;
    pred_bwd8_call<HP, SP, MR, GN, kFL, PG>(W, gargs(mg), stash, readout_div, phase == 2 ? 1 : 0, gnode);
    if (dz_ext != nullptr) {  // + the target's direct dependence on z (callback launches are never packed: slot n = node n)
      for (int e = tid; e < mg.N * mg.D; e += kThreads) sGrad[e] += dz_ext[e];
      __syncthreads();
    }
    w8::guidance_apply(mg, sZ, sGrad, sMean, sigma, tid);
#endif
  }
  template <int HP>
  __device__ __forceinline__ static void pred_entry(const PredDev& W, const Graph& mg, float* net, float* sZ, float* sGrad,
                                                    float* sTmp, float* sMean, float t_val, const float* dpred, bool want_grad,
                                                    float* pred_out, float readout_div, float* stash, int tid STAMP_DECL, float* gnode) {
    w8::predictor_entry<HP, SP, MR, GN, PG>(W, mg, net, sZ, sGrad, sTmp, sMean, t_val, dpred, want_grad, pred_out, readout_div, stash, mg.pubx,
                            mg.pub_ch, tid STAMP_ARGS, gnode);
  }
};
using V8 = V8T<0>;
using V8S = V8T<1>;   // split operands, full weight ring
using V8H = V8T<2>;   // split operands, half ring (two trips per K chunk)

__host__ __device__ inline int common_floats_base(int N, int D) { return 3 * align16(N * D) + align16(N) + 16; }

// V4 -> 256 threads (one wave per SIMD, up to 512 registers); V8 -> 512 threads = two waves per SIMD: the register
// allocator is held to 256 VGPR + AGPR per lane.  P stays a by-value kernel argument (SGPR-resident).
template <class V, int HPE, int HPP>
__global__ __launch_bounds__(V::kThreads) void sampler_kernel_v(const KParams P) {
  constexpr int kThreads = V::kThreads;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if GAUDI_STATIC_PRIO
  // experiment (MI355X_MICROARCH.md, two waves per SIMD, item 4): the second-dispatched half of an 8-wave workgroup loses
  // every issue arbitration by age; one static priority step for it, no per-phase flips
  if (V::kThreads == 512 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int b = P.order[blockIdx.x];
  const int N = P.N, D = 3 + P.F;

  // ---- carve the common region
  float* base = smem;
  float* sZ = base; base += align16(N * D);
  float* sEps = base; base += align16(N * D);
  float* sNz = base; base += align16(N * D);
  float* sMask = base; base += align16(N);
  float* sMean = base; base += 16;          // [component k][4]: masked means of the x columns (d < 3), clip coefficient (3)
  int* sRow = (int*)base; base += align16(N);
  int* sCmol = (int*)base; base += 16;
  for (int i = tid; i < N; i += kThreads) {
    sMask[i] = P.node_mask[b * N + i];
    sRow[i] = P.rowmap != nullptr ? P.rowmap[(size_t)b * N + i] : b * N + i;  // not packed: slot i = node i of molecule b (NR = N)
  }
  if (tid < kMaxComp) sCmol[tid] = P.rowmap != nullptr ? P.compmol[(size_t)b * kMaxComp + tid] : b;
  if (tid == kMaxComp) sCmol[kMaxComp] = P.rowmap != nullptr ? P.ncomp[b] : 1;
  typename V::Graph mg;
  float* net = V::load_graph(P, b, base, sMask, mg, tid, wave);
  __syncthreads();
  const int ncomp = __builtin_amdgcn_readfirstlane(sCmol[kMaxComp]);
  V::set_rows(mg, sRow, ncomp, P.NR);

  // locals (not references into the kernarg struct) so nothing forces P onto the stack
  const float* const noise_p = P.noise;
  const long long draw_stride = P.draw_stride, sample_offset = P.sample_offset, fix_key = P.fix_key;
  const int draw_base = P.draw_base, fix_noise = P.fix_noise, NR = P.NR;
  const unsigned long long seed = P.seed;
  // raw N(0,1) draw `draw`, element (slot n, column d): keyed by the molecule's GLOBAL sample index and the node's index
  // inside its own molecule, so a molecule's noise does not depend on which workgroup (or which slot) holds it
  auto raw_noise = [=](int draw, int n, int d) -> float {
    const int w = sRow[n];
    if (w < 0) return 0.f;
    const int row = row_of(w), local = (row % NR) * D + d;
    if (noise_p != nullptr)
      return noise_p[(size_t)(draw - draw_base) * draw_stride + (fix_noise ? (size_t)local : (size_t)row * D + d)];
    const uint64_t gsample = (uint64_t)(fix_noise ? fix_key : sample_offset + sCmol[comp_of(w)]);
    const f4 v = philox_normal4(seed, gsample, (uint32_t)draw, (uint32_t)(local >> 2));
    return v[local & 3];
  };
  // masked mean over the nodes of component k of column d (<3) of an LDS [N][D] array -> sMean[4 k + d]
  auto col_means = [=](const float* a) {
    if (tid < 3 * ncomp) {
      const int k = tid / 3, d = tid % 3;
      float s = 0.f, cnt = 0.f;
      for (int n = 0; n < N; ++n)
        if (comp_of(sRow[n]) == k) { s += a[n * D + d]; cnt += sMask[n]; }
      sMean[4 * k + d] = s / fmaxf(cnt, 1.0f);
    }
  };
  auto mean_of = [=](int n, int d) { return sRow[n] < 0 ? 0.f : sMean[4 * comp_of(sRow[n]) + d]; };
  // sNz <- sample_combined_position_feature_noise (en_diffusion.py:937-956) from raw draw `draw`
  auto combined_noise = [=](int draw, float std) {
    for (int e = tid; e < N * D; e += kThreads) sNz[e] = raw_noise(draw, e / D, e % D) * std * sMask[e / D];
    __syncthreads();
    col_means(sNz);
    __syncthreads();
    for (int e = tid; e < N * 3; e += kThreads) {
      const int n = e / 3, d = e % 3;
      sNz[n * D + d] = sNz[n * D + d] - mean_of(n, d) * sMask[n];
    }
    __syncthreads();
  };
  // global element of (slot n, column d) in a [rows][W] array, or -1 for an empty slot
  auto gidx = [=](int n, int d, int W) -> long long { return sRow[n] < 0 ? -1LL : (long long)row_of(sRow[n]) * W + d; };

  const int mode = P.mode;
  if (P.clock_out != nullptr && blockIdx.x == 0 && tid == 0) {
    P.clock_out[0] = __builtin_readcyclecounter();
    P.clock_out[1] = __builtin_amdgcn_s_memrealtime();
  }
#ifdef GAUDI_STAMPS
  Stamps g_stamps;
  g_stamps.init();
  const bool stamps_on = P.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (mode == MODE_SAMPLE && P.do_init) {
    __syncthreads();
    combined_noise(0, P.std0);
    for (int e = tid; e < N * D; e += kThreads) sZ[e] = sNz[e];
  } else if (P.alpha_sigma != nullptr) {
    // sample_edm_t (cond_prediction/train_cond_predictor.py:47-61): z_t = alpha_t * normalize([x | h]) + sigma_t * eps
    __syncthreads();
    combined_noise(draw_base, 1.0f);
    const float nv0 = P.nv0, nv1 = P.nv1;
    for (int e = tid; e < N * D; e += kThreads) {
      const int n = e / D, d = e % D;
      const long long gi = gidx(n, d, D);
      const int mol = gi < 0 ? 0 : (int)(gi / ((long long)NR * D));
      const float a_t = P.alpha_sigma[2 * mol], s_t = P.alpha_sigma[2 * mol + 1];
      const float raw = gi < 0 ? 0.f : P.z_in[gi];
      const float xh = d < 3 ? raw / nv0 : (raw - 0.0f) / nv1 * sMask[n];  // en_diffusion.py:384-392
      sZ[e] = gi < 0 ? 0.f : a_t * xh + s_t * sNz[e];
    }
    if (P.zt_out != nullptr) {
      __syncthreads();
      for (int e = tid; e < N * D; e += kThreads) {
        const long long gi = gidx(e / D, e % D, D);
        if (gi >= 0) P.zt_out[gi] = sZ[e];
      }
    }
  } else {
    for (int e = tid; e < N * D; e += kThreads) {
      const long long gi = gidx(e / D, e % D, D);
      sZ[e] = gi < 0 ? 0.f : P.z_in[gi];
    }
  }
  __syncthreads();
  // z of every mapped node -> z_out
  auto store_z = [=](const float* src) {
    for (int e = tid; e < N * D; e += kThreads) {
      const long long gi = gidx(e / D, e % D, D);
      if (gi >= 0) P.z_out[gi] = src[e];
    }
  };

  if constexpr (HPE > 0) {
    if (mode == MODE_PHI || mode == MODE_SAMPLE) {
      const EdmDev edm = P.edm;
      const int guided = P.guided, T = P.T, s_hi = P.s_hi;
      const int n_steps = mode == MODE_SAMPLE ? (s_hi - P.s_lo + 1) : 0;
      const int n_pass = mode == MODE_PHI ? 1 : n_steps + (P.do_decode ? 1 : 0);
      int nan_local = 0;
      // one EDM evaluation per pass: reverse steps s_hi..s_lo, then (optionally) the decode pass
      for (int pass = 0; pass < n_pass; ++pass) {
        const bool is_step = mode == MODE_SAMPLE && pass < n_steps;
        const int s = s_hi - pass;
        f4 cf = splat(0.f);
        if (is_step) cf = *(const f4*)(P.coef + 4 * s);
        const float t_val = mode == MODE_PHI ? P.t_in[b] : cf[3];  // decode: t = 0
        if (is_step == false && mode == MODE_SAMPLE) store_z(sZ);  // z_0 is final: publish it
        const int split = P.split;
        if (split != 2)
          V::template edm<HPE>(edm, mg, net, sZ, sEps, sMean, t_val, tid STAMP_ARGS,
                               V::kGlobalNodes ? P.gnode + (size_t)blockIdx.x * P.gnode_stride : nullptr);
        if (mode == MODE_PHI) {
          store_z(sEps);
        } else if (is_step) {
          if (split != 2) {
            // ---- z_s = z_t/alpha_ts - c*eps + sigma*noise ; x part mean-removed (en_diffusion.py:831-852)
            combined_noise(T - s, 1.0f);
            STAMP(ST_X2);
            for (int e = tid; e < N * D; e += kThreads) {
              float ep = sEps[e];
              if (guided) {  // eps_t.nan_to_num(0.)  (en_diffusion.py:881)
                if (ep != ep) { ep = 0.f; ++nan_local; }
                ep = fminf(fmaxf(ep, -3.4028234663852886e38f), 3.4028234663852886e38f);
              }
              const float mu = sZ[e] / cf[0] - cf[1] * ep;
              sZ[e] = mu + cf[2] * sNz[e];
            }
            __syncthreads();
            STAMP(ST_UPDATE);
          }
          if constexpr (HPP > 0) {
            if (guided) {
              // guidance (en_diffusion.py:899-920): predictor at (z_s, t), clip, project, apply
              const float t_step = cf[3], sigma_step = cf[2];
              V::template guide<HPP>(P.pred, mg, net, sZ, sEps /* grad */, sNz /* scratch */, sMean, t_step, sigma_step,
                                   P.target_w, P.scale, split == 1 ? P.pred_out + (size_t)b * P.pred.K : nullptr,
                                   P.readout_div, P.stash + (size_t)b * P.stash_stride, tid STAMP_ARGS, split,
                                   split == 2 ? P.dpred_in + (size_t)b * P.pred.K : nullptr,
                                   V::kGlobalNodes ? P.gnode + (size_t)blockIdx.x * P.gnode_stride : nullptr,
                                   split == 2 && P.dz_in != nullptr ? P.dz_in + (size_t)b * N * D : nullptr);
            }
          }
          if (split == 1) break;  // phase A ends before the projection: phase B resumes from this z_s
          col_means(sZ);
          __syncthreads();
          for (int e = tid; e < N * 3; e += kThreads) {
            const int n = e / 3, d = e % 3;
            sZ[n * D + d] = sZ[n * D + d] - mean_of(n, d) * sMask[n];
          }
          __syncthreads();
          STAMP(ST_X3);
          if (guided) {
            // `if torch.isnan(zs).any(): zs = zs.nan_to_num(0.)` (en_diffusion.py:933-934): NaN -> 0 and +-inf -> +-FLT_MAX.
            // The reference tests the whole batch; a workgroup sees one molecule (or a few, as components), so the trigger
            // here is "a NaN in THIS molecule" (differs only for a molecule that holds an inf but no NaN while another
            // molecule holds a NaN).
            int bad = 0, badmask = 0;
            for (int e = tid; e < N * D; e += kThreads)
              if (sZ[e] != sZ[e]) { ++bad; badmask |= 1 << comp_of(sRow[e / D]); }
            nan_local += bad;
            badmask = block_or_bits(badmask & ((1 << kMaxComp) - 1), ncomp);
            if (badmask) {
              for (int e = tid; e < N * D; e += kThreads) {
                float v = sZ[e];
                if ((badmask >> comp_of(sRow[e / D])) & 1) {
                  v = v != v ? 0.f : fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
                  sZ[e] = v;
                }
              }
              __syncthreads();
            }
          }
          STAMP(ST_UPDATE);
          if (P.chain_out != nullptr) {
            // sample_chain (en_diffusion.py:1145-1161): frame (s*K)//T receives unnormalize_z(z_s); a later (smaller) s
            // mapping to the same frame overwrites it, so only the last writer of each frame stores.
            const int K = P.keep_frames;
            const int idx = (int)(((long long)s * K) / T);
            if (s == 0 || (int)(((long long)(s - 1) * K) / T) != idx) {
              float* dst = P.chain_out + (size_t)idx * P.B * N * D;  // (never packed: B = molecules)
              for (int e = tid; e < N * D; e += kThreads) {
                const int n = e / D, d = e % D;
                const long long gi = gidx(n, d, D);
                if (gi >= 0) dst[gi] = d < 3 ? sZ[e] * P.nv0 : (sZ[e] * P.nv1 + 0.0f) * sMask[n];
              }
            }
          }
        } else {
          // ---- decode: sample_p_xh_given_z0 (en_diffusion.py:533-560) + unnormalize (:406-415)
          combined_noise(T + 1, 1.0f);
          const float inv_a = 1.0f / P.alpha0;
          const float sigma0 = P.sigma0, sigma_x = P.sigma_x, nv0 = P.nv0, nv1 = P.nv1;
          const int F = P.F;
          for (int e = tid; e < N * 3; e += kThreads) {
            const int n = e / 3, d = e % 3;
            const float mu = inv_a * (sZ[n * D + d] - sigma0 * sEps[n * D + d]);
            const long long gi = gidx(n, d, 3);
            if (gi >= 0) P.x_out[gi] = (mu + sigma_x * sNz[n * D + d]) * nv0;
          }
          for (int n = tid; n < N; n += kThreads) {
            if (sRow[n] < 0) continue;
            int best = 0;
            float bv = (sZ[n * D + 3] * nv1 + 0.0f) * sMask[n];
            for (int k = 1; k < F; ++k) {
              const float v = (sZ[n * D + 3 + k] * nv1 + 0.0f) * sMask[n];
              if (v > bv) { bv = v; best = k; }
            }
            for (int k = 0; k < F; ++k) P.h_out[(size_t)row_of(sRow[n]) * F + k] = (k == best ? 1.0f : 0.0f) * sMask[n];
          }
        }
      }
      if (mode == MODE_SAMPLE && !P.do_decode) store_z(sZ);
#ifdef GAUDI_STAMPS
      if (stamps_on) {
        for (int i = 0; i < ST_N; ++i) P.stamps[i] = g_stamps.acc[i];
        P.stamps[30] = __builtin_readcyclecounter() - clk0;         // shader clock
        P.stamps[31] = __builtin_amdgcn_s_memrealtime() - rt0;      // constant 100 MHz
      }
#endif
      if (P.clock_out != nullptr && blockIdx.x == 0 && tid == 0) {
        P.clock_out[2] = __builtin_readcyclecounter();
        P.clock_out[3] = __builtin_amdgcn_s_memrealtime();
      }
      if (nan_local) atomicAdd(P.nan_count, nan_local);
      return;
    }
  }
  if constexpr (HPP > 0 && HPE == 0) {
    if (P.mode == MODE_GUIDE) {
      // Second half of a guided reverse step as its own launch (the V4G path: its fused EDM + predictor instantiation at the
      // default widths sits on the register cliff, so large molecules run "EDM-only kernel with split = 1, then this"):
      // z_in = z_s before guidance (en_diffusion.py:897) -> guidance update, projection, NaN scrub (:899-934) -> z_out.
      // P.split = 1 / 2: the two halves of this launch around a host callback (gaudi_sample_cb on large molecules): 1 = predictor
      // forward only, pred -> pred_out, z untouched; 2 = reverse pass with dT/dpred = dpred_in, then the update below.
      const f4 cf = *(const f4*)(P.coef + 4 * P.s_hi);
      const int gsplit = P.split;
      V::template guide<HPP>(P.pred, mg, net, sZ, sEps /* grad */, sNz /* scratch */, sMean, cf[3], cf[2], P.target_w, P.scale,
                             gsplit == 1 ? P.pred_out + (size_t)b * P.pred.K : nullptr,
                             P.readout_div, P.stash + (size_t)b * P.stash_stride, tid STAMP_ARGS, gsplit,
                             gsplit == 2 ? P.dpred_in + (size_t)b * P.pred.K : nullptr,
                             V::kGlobalNodes ? P.gnode + (size_t)blockIdx.x * P.gnode_stride : nullptr,
                             gsplit == 2 && P.dz_in != nullptr ? P.dz_in + (size_t)b * N * D : nullptr);
      if (gsplit == 1) return;
      col_means(sZ);
      __syncthreads();
      for (int e = tid; e < N * 3; e += kThreads) {
        const int n = e / 3, d = e % 3;
        sZ[n * D + d] = sZ[n * D + d] - mean_of(n, d) * sMask[n];
      }
      __syncthreads();
      int bad = 0;
      for (int e = tid; e < N * D; e += kThreads) bad += sZ[e] != sZ[e];
      if (__syncthreads_or(bad)) {
        for (int e = tid; e < N * D; e += kThreads) {
          float v = sZ[e];
          sZ[e] = v != v ? 0.f : fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
        }
        __syncthreads();
      }
      if (bad) atomicAdd(P.nan_count, bad);
      store_z(sZ);
    }
    // unit-test modes live in the predictor-only kernels
    if (P.mode == MODE_PRED_FWD || P.mode == MODE_PRED_GRAD) {
      const float* dp = P.dpred_in ? P.dpred_in + (size_t)b * P.pred.K : nullptr;
      V::template pred_entry<HPP>(P.pred, mg, net, sZ, sEps, sNz, sMean, P.t_in[b], dp, P.mode == MODE_PRED_GRAD,
                           P.pred_out + (size_t)b * P.pred.K, P.readout_div, P.stash + (size_t)b * P.stash_stride, tid STAMP_ARGS,
                           V::kGlobalNodes ? P.gnode + (size_t)blockIdx.x * P.gnode_stride : nullptr);
      if (P.mode == MODE_PRED_GRAD) store_z(sEps);
    }
  }
}

typedef void (*sampler_fn)(const KParams);
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel = &sampler_kernel_v<V4, HPE, HPP>;
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel_g = &sampler_kernel_v<V4G, HPE, HPP>;
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel_se = &sampler_kernel_v<V4S, HPE, HPP>;
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel_gse = &sampler_kernel_v<V4GS, HPE, HPP>;
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8 = &sampler_kernel_v<V8, HPE, HPP>;
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8s = &sampler_kernel_v<V8S, HPE, HPP>;
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8s2 = &sampler_kernel_v<V8T<1, false, false, true>, HPE, HPP>;  // (FR: kern8s2_*.hip)
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8h = &sampler_kernel_v<V8H, HPE, HPP>;
// ... whose predictor runs several rounds of edge tiles (kern8m_*.hip): SP = 0 / 1 / 2 as above
template <int SP, int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8m = &sampler_kernel_v<V8T<SP, true>, HPE, HPP>;
// ... and with the node buffers in global memory (kern8g_*.hip: molecules beyond the LDS limit; split edge GEMMs, full ring,
// several rounds of edge tiles in the predictor)
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8g = &sampler_kernel_v<V8T<1, true, 1>, HPE, HPP>;
// ... wide groups on the FULL ring: several rounds of edge tiles, the predictor's fifth node buffer in the global scratch (kern8mp_*.hip)
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8mp = &sampler_kernel_v<V8T<1, true, 0, false, true>, HPE, HPP>;
// ... of which P and Q stay in LDS (kern8gp_*.hip: taken where that plan fits)
template <int HPE, int HPP>
inline constexpr sampler_fn sampler_kernel8gp = &sampler_kernel_v<V8T<1, true, 2>, HPE, HPP>;

}  // namespace gaudi
