// gaudi_hip.hip -- host side of libgaudi_hip.so: the C ABI declared in include/gaudi_hip.h.
// Builds the noise-schedule tables, packs reference-format checkpoints into the kernels' tile
// layout, turns (node_mask, edge_mask) into per-wave edge lists and launches the per-molecule
// persistent kernel (sampler_kernel.h) on the handle's own HIP stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <queue>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gaudi_hip.h"
#include "sampler_kernel.h"

using namespace gaudi;

// -------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return (T*)p; }
};

// pinned host memory (the per-step [B,K] exchange of gaudi_sample_cb: a pageable hipMemcpyAsync is staged and blocks)
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return (T*)p; }
};

struct gaudi_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::string warn;  // the last non-fatal condition a caller should know about (gaudi_last_warning)
  bool has_edm = false, has_pred = false;
  gaudi_edm_config ecfg{};
  gaudi_pred_config pcfg{};
  int HPE = 0, HPP = 0;
  DevBuf edm_w, pred_w, coef_d, edm_w4, pred_w4;  // *_w4: row-major tiles for the 4-wave fallback of an 8-wave handle
  DevBuf edm_ws, pred_ws;                         // fp16-pair images of the edge-GEMM matrices (w8_split.h)
  size_t edm_w_bytes = 0, pred_w_bytes = 0, edm_ws_bytes = 0, pred_ws_bytes = 0;
  float edm_hinv = 0.f, pred_hinv = 0.f;  // 2^-s of the fp16-pair node images inside *_ws (w8_nodes_f16.h); 0: the weight set
                                          // was refused (node_scale) -- its calls run the fp32-instruction kernels
  std::vector<float> gamma, coef;
  // per-call workspaces
  DevBuf d_mask, d_order, d_edges, d_emask, d_npairs, d_seg, d_zin, d_zout, d_t, d_x, d_h, d_noise, d_nan, d_dpred,
      d_pred, d_tw, d_stash, d_chain, d_sx, d_stype, d_sn, d_sflags, d_sdist, d_sadj, d_saux, d_stab, d_as, d_ncols, d_soff,
      d_sidx, d_gnode, d_rowmap, d_compmol, d_ncomp, d_clock;
  PinBuf p_pred, p_dpred;     // gaudi_sample_cb: pred [B,K] device -> host, dT/dpred [B,K] host -> device, once per step
  PinBuf p_z, p_dz;           // gaudi_sample_cbz: z_s [B,N,D] device -> host, scale * dT/dz host -> device
  DevBuf d_dz;
  hipEvent_t cb_event = nullptr;
  int steps_per_launch = 25;
  int variant = 8;            // 8 = two waves per SIMD (sampler_kernel8, default), 4 = one wave per SIMD (GAUDI_WAVES=4)
  int run_variant = 4;        // what the CURRENT call runs on (an 8-wave handle falls back to 4 waves for graphs that do not fit)
  bool split = true;          // 8-wave kernels: GEMMs on the fp16 matrix pipe with operands split into fp16 pairs (GAUDI_EDGE_MATH=fp32: off)
  int run_split = 0;          // ... and how the CURRENT call uses them: 1 = full weight ring, 2 = half ring, 0 = fp32 instructions
  bool run_gn = false;        // the CURRENT call runs on the 4-wave kernels with node buffers in global memory (large molecules)
  bool run_two = false;       // ... and its guided steps are two launches (denoiser-only kernel, then predictor-only kernel): no fused instantiation
  int run_gn8 = 0;            // ... on the 8-wave kernels with node buffers in global memory (V8G, round 4): 1 all five, 2 P / Q in LDS
  bool run_pg = false;        // ... a wide-group launch on the FULL ring with the predictor's fifth node buffer in the global scratch (kern8mp_*.hip)
  bool wide_full = true;      // GAUDI_WIDE_FULL=0: wide groups that do not fit the full ring run on the half ring (round 5)
  bool gn8_pq = true;         // GAUDI_GN8_PQ=0: never the P / Q-in-LDS form (kern8gp_*.hip)
  bool gn8 = true;            // GAUDI_GN8=0: molecules beyond the LDS limit go to the 4-wave V4G kernels, as in round 3
  bool force_gn8 = false;     // GAUDI_FORCE_GN8=1: V8G whenever it can run (test knob)
  bool pack = true;           // several small molecules per workgroup in sampling calls (GAUDI_PACK=0: off)
  int pairs_cap = -1;         // GAUDI_PAIRS_CAP: two-molecule groups of a wide launch (-1: chosen by wide_plan below; experiments)
  int pairs = 1;              // wide groups (more node slots than a molecule has, two rounds of edge tiles: e.g. two 11-ring cata
                              // molecules per workgroup): 0 = never, 1 = when the batch holds at least two molecules per CU and the
                              // classic packing shares next to nothing (default since round 5: the fp16-pair node GEMMs made the
                              // shared weight stream worth the second round), 2 = always (GAUDI_PAIRS)
  int num_cus = 256;
  int run_nslots = 0;         // node slots per workgroup of the current call (= N unless the call runs wide groups)
  bool pred_rounds = true;    // GAUDI_PRED_ROUNDS=0: guided calls with more than 128 edge slots go to the 4-wave kernels
  bool pack_now = false;      // set by run_chain around stage_graph: this call may pack
  int run_groups = 0;         // workgroups of the CURRENT call (= molecules unless packed)
  bool force_mr = false;      // GAUDI_FORCE_MR
  bool run_mr = false;        // the current call runs the 8-wave kernels whose predictor takes several rounds of edge tiles
  bool force_gn = false;      // GAUDI_FORCE_GN=1 at gaudi_create: use them whenever they exist (test knob)
  bool fix_noise = false;     // en_diffusion.py:562-566: one raw draw per call, broadcast over the batch
  long long fix_key = 0;      // global sample index whose Philox stream is shared
  int readout_n = 0;  // padded N the predictor readout divides by (0 = the call's N)
  // Plan hint (gaudi_set_plan_hint): the kernel family and the edge-GEMM arithmetic of a call follow from batch-wide maxima
  // (edge slots, a node's live edges).  A shard of a larger logical batch plans with the WHOLE batch's figures, so that a
  // molecule's rounding does not depend on where the batch was cut.  call_* = the same, set by gaudi_sample for its own
  // sub-batches.
  int plan_min_slots = 0, plan_force_waves = 0;
  int call_min_slots = 0, call_force_waves = 0;
  bool call_cut = false;      // gaudi_sample cut this request into sub-batches
  // Per-molecule kernel family (round 6): gaudi_sample sorts a request whose padded N is beyond the resident kernels' LDS limit into
  // molecules that fit those kernels on their OWN (few enough live nodes and edge tiles) and the rest (V8G), and runs the two
  // buckets one after the other.  call_narrow = node slots per workgroup of the first bucket's packed launch (< N); call_molmap =
  // the bucket's molecule -> index in the request (the Philox key is the molecule's global sample index).
  int call_narrow = 0;
  const int32_t* call_molmap = nullptr;
  bool keep_h = true;         // GAUDI_KEEP_H=0: every node GEMM that reads h splits it again (round 5)
  int run_hk = 0;             // floats of the kept split copy of h behind the current call's LDS plan (0: none)
  bool gn8_pack = true;       // GAUDI_GN8_PACK=0: V8G launches keep a molecule's nodes where the masks have them (round 5)
  bool family_split = false;  // GAUDI_FAMILY_SPLIT=1: per-molecule kernel family (below).  Off by default: the two buckets run as two launches
                              // per 25 steps on one stream, each with its own tail -- c4x 76.8 against 84.3 mol/s in one family
  int last_split_resident = 0;  // molecules of the last gaudi_sample call that ran on the resident kernels beside a V8G bucket
  // profiling: launches are bracketed by HIP events on the handle's stream.  A small window of pending pairs is kept;
  // older pairs are folded into running sums and their events recycled (a T = 1000 callback chain makes 2001 launches).
  bool prof = false;
  struct EventLog {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, spare;
    double ms = 0.0;
    long long n = 0;
    static constexpr size_t kWindow = 32;
    hipError_t fold(size_t keep) {
      while (pending.size() > keep) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(pending.front().second);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, pending.front().first, pending.front().second);
        if (e != hipSuccess) return e;
        ms += t;
        ++n;
        spare.push_back(pending.front());
        pending.erase(pending.begin());
      }
      return hipSuccess;
    }
    hipError_t begin(hipStream_t st, std::pair<hipEvent_t, hipEvent_t>& ev) {
      hipError_t e = fold(kWindow - 1);
      if (e != hipSuccess) return e;
      if (!spare.empty()) {
        ev = spare.back();
        spare.pop_back();
      } else {
        if ((e = hipEventCreate(&ev.first)) != hipSuccess) return e;
        if ((e = hipEventCreate(&ev.second)) != hipSuccess) return e;
      }
      return hipEventRecord(ev.first, st);
    }
    hipError_t end(hipStream_t st, const std::pair<hipEvent_t, hipEvent_t>& ev) {
      pending.push_back(ev);
      return hipEventRecord(ev.second, st);
    }
    void reset(bool destroy) {
      for (auto& p : pending) spare.push_back(p);
      pending.clear();
      if (destroy) {
        for (auto& p : spare) {
          (void)hipEventDestroy(p.first);
          (void)hipEventDestroy(p.second);
        }
        spare.clear();
      }
      ms = 0.0;
      n = 0;
    }
  } prof_log, stab_log;
  long long prof_steps = 0;
#ifdef GAUDI_STAMPS
  DevBuf d_stamps;
  unsigned long long stamp_acc[32] = {0};
#endif
};

#define HIPCHECK(h, call)                                                                         \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                              \
      return GAUDI_E_HIP;                                                                         \
    }                                                                                             \
  } while (0)

static int fail(gaudi_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

// -------------------------------------------------------------------------------------------------
// noise schedule: PredefinedNoiseSchedule / polynomial_schedule / clip_noise_schedule
// (edm/equivariant_diffusion/en_diffusion.py:32-61, 191-218), float64 like the reference's numpy.
static std::vector<float> make_gamma(int T, double power, double precision) {
  const int steps = T + 1;
  if (power <= 0.0) {
    // the 'cosine' schedule (cosine_beta_schedule, en_diffusion.py:64-81; encoded as noise_power = 0): alphas_cumprod of
    // T + 2 points of cos^2, betas clipped to [0, 0.999]; no precision blend
    const int n = T + 2;
    const double s_ = 0.008, pi = 3.14159265358979323846;
    std::vector<double> ac(n);
    for (int i = 0; i < n; ++i) {
      const double x = (i == n - 1) ? (double)n : i * ((double)n / (double)(n - 1));  // np.linspace(0, n, n)
      const double c = std::cos(((x / n) + s_) / (1.0 + s_) * pi * 0.5);
      ac[i] = c * c;
    }
    const double a0 = ac[0];
    for (double& v : ac) v /= a0;
    std::vector<float> g(steps);
    double cum = 1.0;
    for (int i = 0; i < steps; ++i) {
      double beta = 1.0 - ac[i + 1] / ac[i];
      beta = std::min(std::max(beta, 0.0), 0.999);
      cum *= 1.0 - beta;
      g[i] = (float)(-(std::log(cum) - std::log(1.0 - cum)));
    }
    return g;
  }
  std::vector<double> a2(steps);
  const double step = (double)steps / (double)(steps - 1);  // np.linspace(0, steps, steps)
  for (int i = 0; i < steps; ++i) {
    double x = (i == steps - 1) ? (double)steps : i * step;
    double v = 1.0 - std::pow(x / steps, power);
    a2[i] = v * v;
  }
  // clip_noise_schedule: ratios alpha2[i]/alpha2[i-1] clipped to [1e-3, 1], cumulative product
  std::vector<double> out(steps);
  double prev = 1.0, cum = 1.0;
  for (int i = 0; i < steps; ++i) {
    double r = a2[i] / prev;
    r = std::min(std::max(r, 0.001), 1.0);
    cum *= r;
    out[i] = cum;
    prev = a2[i];
  }
  std::vector<float> g(steps);
  const double pr = 1.0 - 2.0 * precision;
  for (int i = 0; i < steps; ++i) {
    double a = pr * out[i] + precision;
    g[i] = (float)(-(std::log(a) - std::log(1.0 - a)));
  }
  return g;
}

static inline float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
static inline float logsigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
static inline float sigmoid_host(float x) { return 1.0f / (1.0f + expf(-x)); }

// en_diffusion.py:433-457 (sigma_and_alpha_t_given_s), :365-373 (sigma), :843-849 (mu / sigma of p(z_s|z_t))
static void make_coef(const std::vector<float>& g, int T, std::vector<float>& coef) {
  coef.resize((size_t)T * 4);
  for (int s = 0; s < T; ++s) {
    const float gs = g[s], gt = g[s + 1];
    const float sigma2 = -expm1f(softplus_f(gs) - softplus_f(gt));
    const float alpha_ts = expf(0.5f * (logsigmoid_f(-gt) - logsigmoid_f(-gs)));
    const float sigma_ts = sqrtf(sigma2);
    const float sigma_s = sqrtf(sigmoid_host(gs)), sigma_t = sqrtf(sigmoid_host(gt));
    coef[4 * s + 0] = alpha_ts;
    coef[4 * s + 1] = sigma2 / alpha_ts / sigma_t;
    coef[4 * s + 2] = sigma_ts * sigma_s / sigma_t;
    coef[4 * s + 3] = (float)(s + 1) / (float)T;
  }
}

// -------------------------------------------------------------------------------------------------
// checkpoint packing
struct Tensors {
  std::map<std::string, std::pair<const float*, int64_t>> m;
  std::string missing;
  const float* get(const std::string& k, int64_t numel) {
    auto it = m.find(k);
    if (it == m.end() || it->second.second != numel) {
      if (missing.empty())
        missing = k + (it == m.end() ? " (absent)" : " (numel " + std::to_string(it->second.second) + " != " +
                                                         std::to_string(numel) + ")");
      return nullptr;
    }
    return it->second.first;
  }
  // the tensor if it is there with that size (no complaint: get() of the same tensor reports what is wrong)
  const float* peek(const std::string& k, int64_t numel) const {
    auto it = m.find(k);
    return it == m.end() || it->second.second != numel ? nullptr : it->second.first;
  }
};

// Packing context of one checkpoint load (a value passed down -- no file-scope state: two handles may load concurrently).
//   lane_linear (8-wave kernels): inside a 16x16 tile float4 index L = (k%16/4)*16 + o%16 holds W[o][k .. k+3], i.e. lane L
//     of a wave reads the 16 bytes at offset 16 L whether the tile comes from L2 or from the LDS weight ring; otherwise the
//     tile is row-major (4-wave kernels)
//   ktail (8-wave kernels, H % 16 == 4): the last K chunk carries its 4 valid inputs on element 0 of the four lane groups
//   wbase / ws: when a split buffer is being filled, the fp32 buffer's base and the split image (w8_split.h) whose float
//     offsets are twice the fp32 ones
struct PackMode {
  bool lane_linear = false;
  bool ktail = false;
  float* wbase = nullptr;
  std::vector<float>* ws = nullptr;
  float hscale = 0.f;   // 2^s of the network's fp16-pair node images (w8_nodes_f16.h); 0: no such images
  bool hktail = false;  // ... their K tail (H % 16 == 4)
  PackMode with_ktail(bool kt) const {
    PackMode m = *this;
    m.ktail = kt;
    return m;
  }
};
static bool has_ktail(int H, int HP) { return HP - H == 12; }
// W[o][col0 + k] (row stride ldw), o,k < H  ->  tile-packed [HP/16][HP/16][16][16]: dst[((k/16*T + o/16)*16 + o%16)*16 + k%16]
static void pack_matrix(const PackMode& pm, float* dst, const float* W, int H, int ldw, int col0, int HP, bool transpose = false) {
  const int T = HP / 16;
  for (int o = 0; o < H; ++o)
    for (int k = 0; k < H; ++k) {
      const float v = transpose ? W[(size_t)k * ldw + col0 + o] : W[(size_t)o * ldw + col0 + k];
      const size_t tile = ((size_t)(k / 16) * T + o / 16) * 256;
      size_t in = pm.lane_linear ? (size_t)(((k % 16) / 4) * 16 + o % 16) * 4 + k % 4 : (size_t)(o % 16) * 16 + k % 16;
      if (pm.lane_linear && pm.ktail && k / 16 == T - 1) in = (size_t)((k % 16) * 16 + o % 16) * 4;  // k % 16 < 4 here
      dst[tile + in] = v;
    }
}
// fp16 helpers of the weight images (round to nearest even; NaN stays NaN)
static uint16_t f16_bits(float x) {
  const _Float16 hv = (_Float16)x;
  uint16_t u;
  std::memcpy(&u, &hv, 2);
  return u;
}
static float f16_value(uint16_t b) {
  _Float16 hv;
  std::memcpy(&hv, &b, 2);
  return (float)hv;
}
// Split image of an edge-GEMM matrix (w8_split.h, round 5: fp16 pairs): units of 1 KiB ordered [K chunk m][output tile t][piece p];
// lane L = (row L & 15, group g = L >> 4) holds 8 fp16: slots 0-3 = inputs 16(2m) + 4g .. +3, slots 4-7 = inputs 16(2m+1) + 4g .. +3.
// Pieces of w s (s = the network's power-of-two scale, NodeScale below): hi = fp16(w s), lo = fp16(w s - hi), both round-to-
// nearest-even; a NaN weight gives NaN pieces (the product is NaN like the fp32 product), infinite weights are refused by
// NodeScale.  K tail (pm.ktail, odd tile count >= 3 = SplitGeo::kTailOK): the last chunk holds only the tail tile; it is stored
// as T fp32 tiles in pack_matrix's K-tail form, scaled by s (the accumulators run in scaled units), and issued as one fp32
// k-step per tile.
static void pack_matrix_split(const PackMode& pm, float* dst, const float* W, int H, int ldw, int col0, int HP, bool transpose) {
  const int T = HP / 16;
  const bool tail = pm.ktail && (T & 1) && T >= 3;
  const float sc = pm.hscale > 0.f ? pm.hscale : 1.0f;
  uint16_t* d = (uint16_t*)dst;
  for (int o = 0; o < H; ++o)
    for (int k = 0; k < H; ++k) {
      const float v = (transpose ? W[(size_t)k * ldw + col0 + o] : W[(size_t)o * ldw + col0 + k]) * sc;
      const int tile = k / 16, m = tile / 2, g = (k % 16) / 4, e = 4 * (tile & 1) + (k & 3), t = o / 16, L = g * 16 + o % 16;
      if (tail && tile == T - 1) {  // k % 16 < 4 here
        dst[(size_t)(m * T * w8::kPieces + t) * 256 + (size_t)((k % 16) * 16 + o % 16) * 4] = v;
        continue;
      }
      const uint16_t hi = f16_bits(v);
      d[((size_t)((m * T + t) * w8::kPieces + 0) * 64 + L) * 8 + e] = hi;
      d[((size_t)((m * T + t) * w8::kPieces + 1) * 64 + L) * 8 + e] = f16_bits(v - f16_value(hi));
    }
}
// An edge-GEMM matrix: the fp32 tiles (K tail included where the width has one) and, when a split buffer is being filled,
// its split image at twice the float offset
static void pack_edge_matrix(const PackMode& pm, float* dst, const float* W, int H, int ldw, int col0, int HP, bool transpose = false) {
  pack_matrix(pm, dst, W, H, ldw, col0, HP, transpose);
  if (pm.ws != nullptr) pack_matrix_split(pm, pm.ws->data() + 2 * (size_t)(dst - pm.wbase), W, H, ldw, col0, HP, transpose);
}
// fp16-pair image of a node-GEMM matrix (w8_nodes_f16.h): w 2^s = hi + lo 2^-11 with hi = fp16(w 2^s), lo = fp16((w 2^s - hi) 2^11), both
// round-to-nearest-even (NaN stays NaN; the host refuses infinite weights, node_scale below).  Units of 1 KiB ordered
// [K chunk m of 32 inputs][output tile t][piece]; lane L = (row L & 15, group g = L >> 4) holds inputs 32 m + 8 g .. +7.
// An odd tile count leaves a last half chunk of 16 inputs: a trailing block of T x 256 floats, UNSCALED fp32, [tile][k-step q]
// [lane (row, g)] = W[row][16 (T-1) + 4 q + g]: v_mfma_f32_16x16x4_f32 steps (one of them when H % 16 == 4).
static void pack_matrix_f16(float* dst, const float* W, int H, int ldw, int col0, int HP, bool transpose, float scale) {
  const int T = HP / 16;
  uint16_t* d = (uint16_t*)dst;
  for (int o = 0; o < H; ++o)
    for (int k = 0; k < H; ++k) {
      const float v = transpose ? W[(size_t)k * ldw + col0 + o] : W[(size_t)o * ldw + col0 + k];
      const int t = o / 16, i = o % 16;
      if (w8::nh_odd(HP) && k >= 16 * (T - 1)) {  // the last half chunk: fp32, unscaled, [tile][k-step q][lane (row, g)]
        const int kk = k - 16 * (T - 1);
        dst[(size_t)w8::nh_tail_off(HP) + t * 256 + (kk / 4) * 64 + (kk % 4) * 16 + i] = v;
        continue;
      }
      const int m = k / 32, L = ((k % 32) / 8) * 16 + i, e = k % 8;
      const float vs = v * scale;
      const uint16_t hi = f16_bits(vs);
      const uint16_t lo = f16_bits((vs - f16_value(hi)) * 2048.f);
      d[(((size_t)(m * T + t) * 2 + 0) * 64 + L) * 8 + e] = hi;
      d[(((size_t)(m * T + t) * 2 + 1) * 64 + L) * 8 + e] = lo;
    }
}
// One power of two for all matrices of a network (node-level AND edge-level: w8_nodes_f16.h, w8_split.h): the largest finite
// |w| lands in [2^13, 2^14).  An entry keeps 22 significant bits down to 2^-17 of the largest one in the edge images (lo shares
// hi's exponent range: one accumulator) and down to 2^-25 in the node images (lo carries its own exponent), with fp16's
// absolute floor below that.  The form is refused -- the call then runs the fp32-instruction kernels, and the load says so
// (gaudi_last_warning) -- when a matrix holds an infinity (fp16 pieces would turn inf x 0 and inf - inf into NaN where the fp32
// product keeps inf) or when a whole matrix lies below those floors.
struct NodeScale {
  // node-level and edge-level matrices apart: the node images carry an entry with 22 bits down to 2^-25 of the scale's top (lo has
  // its own exponent), the edge images down to 2^-17 (lo shares hi's range)
  float gmax = 0.f, min_node = INFINITY, min_edge = INFINITY;
  bool inf = false;
  void see(const float* W, int rows, int ldw, int col0, int cols, bool edge = false) {
    if (!W) return;
    float m = 0.f;
    for (int o = 0; o < rows; ++o)
      for (int k = 0; k < cols; ++k) {
        const float a = std::fabs(W[(size_t)o * ldw + col0 + k]);
        if (std::isinf(a)) inf = true;
        else if (a == a) m = std::max(m, a);
      }
    gmax = std::max(gmax, m);
    if (m > 0.f) (edge ? min_edge : min_node) = std::min(edge ? min_edge : min_node, m);
  }
  // Round 6: the rule is as narrow as the images allow.  Through round 5 ANY matrix 2^12 below the network's
  // largest entry refused the set (one small matrix: the whole network at 0.55 x the speed, silently).  Now: an edge matrix may lie
  // 2^17 below (its largest entries then still carry 21-22 bits, its error stays at 2^-22 of ITS largest entry -- the norm a GEMM's
  // error is measured in), a node matrix 2^25; and the refusal is loud (gaudi_last_warning).  One exponent per MATRIX instead was
  // built and measured: parity-green, 1.0-1.5 % slower on the headline in three forms (profiles/r06c_per_matrix_scale_ab.txt,
  // tools/experiments/per_matrix_scale.patch).
  bool too_small() const { return min_edge < gmax * 7.62939453125e-6f /* 2^-17 */ || min_node < gmax * 2.98023223876953125e-8f /* 2^-25 */; }
  // -> 2^s (0: refused; why() says which rule)
  float scale() const {
    if (inf || !(gmax > 0.f) || too_small()) return 0.f;
    int ex;
    std::frexp(gmax, &ex);  // gmax = f 2^ex, f in [0.5, 1)
    if (14 - ex > 126 || 14 - ex < -126) return 0.f;  // (the scale and its inverse must be normal numbers: |w| around 2^-112 .. 2^140)
    return std::ldexp(1.f, 14 - ex);
  }
  const char* why() const {
    if (inf) return "a weight matrix holds an infinity";
    if (!(gmax > 0.f)) return "every weight matrix is zero or NaN";
    if (too_small())
      return "the largest entry of some weight matrix lies more than 2^17 (edge-level matrices; 2^25: node-level) below the largest entry of the network";
    return "the largest weight is outside 2^-112 .. 2^140";
  }
};
// A node-GEMM matrix: the fp32 tiles and, when the split buffer is being filled and the network's scale is known, its fp16-pair
// image at twice the float offset (the split buffer's slots of the node matrices)
static void pack_node_matrix(const PackMode& pm, float* dst, const float* W, int H, int ldw, int col0, int HP, bool transpose = false) {
  pack_matrix(pm, dst, W, H, ldw, col0, HP, transpose);
  if (pm.ws != nullptr && pm.hscale > 0.f)
    pack_matrix_f16(pm.ws->data() + 2 * (size_t)(dst - pm.wbase), W, H, ldw, col0, HP, transpose, pm.hscale);
}
static void pack_vec(float* dst, const float* v, int n) { std::memcpy(dst, v, sizeof(float) * n); }
static void pack_col(float* dst, const float* W, int H, int ldw, int col) {
  for (int o = 0; o < H; ++o) dst[o] = W[(size_t)o * ldw + col];
}
// max |W[o][col]| over the rows (NaN if any entry is: the bound it enters must not hide one)
static float col_absmax(const float* W, int H, int ldw, int col) {
  float m = 0.f;
  bool nan = false;
  for (int o = 0; o < H; ++o) {
    const float a = std::fabs(W[(size_t)o * ldw + col]);
    nan |= a != a;
    m = std::max(m, a);
  }
  return nan ? NAN : m;
}

// -------------------------------------------------------------------------------------------------
// graph metadata: live edges of each molecule, receiving nodes dealt to the 4 waves (LPT), each
// wave's edge list sorted by receiving node and padded to 32 ("bucketed by degree").
struct Meta {
  int EW = 32;
  std::vector<int> order, npairs, ncols;
  std::vector<uint32_t> edges, seg;
  std::vector<float> emask;
};

static int build_meta(int B, int N, const float* node_mask, const float* edge_mask, Meta& M, std::string& err) {
  if (N > 255) {
    err = "N > 255 unsupported";
    return GAUDI_E_CAPACITY;
  }
  struct Mol {
    std::vector<std::vector<std::pair<int, float>>> nbr;  // per receiving node: (j, mask)
    int owner[256];
    int total = 0;
  };
  std::vector<Mol> mols(B);
  M.ncols.assign(B, 1);
  int maxlen = 0;
  std::vector<std::vector<int>> wl(B * kWaves);
  for (int b = 0; b < B; ++b) {
    Mol& m = mols[b];
    m.nbr.resize(N);
    // Edges between two MASKED nodes are dropped: the reference leaves the identity block "padded ring <-> its
    // orientation node" unmasked (sampling_edm.py:147-159), but a masked node's features are multiplied by node_mask = 0
    // after every layer and its outputs are masked, and no live node has an edge to it, so those messages reach nothing.
    // ncols = 1 + the last node that is live or has a live edge: node-level GEMMs stop there (hetero batches padded to
    // 20 nodes: molecules of up to 6 rings need one 16-node column tile instead of two).
    int last = 0;
    for (int i = 0; i < N; ++i) {
      const bool li = node_mask == nullptr || node_mask[(size_t)b * N + i] != 0.f;
      if (li) last = i;
      for (int j = 0; j < N; ++j) {
        const float v = edge_mask[((size_t)b * N + i) * N + j];
        const bool lj = node_mask == nullptr || node_mask[(size_t)b * N + j] != 0.f;
        if (v != 0.f && (li || lj)) {
          m.nbr[i].push_back({j, v});
          ++m.total;
          last = std::max(last, std::max(i, j));
        }
      }
    }
    M.ncols[b] = last + 1;
    // longest-processing-time assignment of receiving nodes to waves
    std::vector<int> idx(N);
    for (int i = 0; i < N; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return m.nbr[a].size() > m.nbr[c].size(); });
    int load[kWaves] = {0, 0, 0, 0};
    for (int i : idx) {
      int w = 0;
      for (int k = 1; k < kWaves; ++k)
        if (load[k] < load[w]) w = k;
      m.owner[i] = w;
      load[w] += (int)m.nbr[i].size();
    }
    for (int w = 0; w < kWaves; ++w) maxlen = std::max(maxlen, load[w]);
  }
  M.EW = std::max(32, (maxlen + 31) / 32 * 32);
  if (M.EW > 0x7fff) {
    err = "too many edges per wave";
    return GAUDI_E_CAPACITY;
  }
  const int EW = M.EW;
  M.edges.assign((size_t)B * kWaves * EW, 0);
  M.emask.assign((size_t)B * kWaves * EW, 0.f);
  M.npairs.assign((size_t)B * kWaves, 0);
  M.seg.assign((size_t)B * N, 0);
  for (int b = 0; b < B; ++b) {
    Mol& m = mols[b];
    int fill[kWaves] = {0, 0, 0, 0};
    int last_i[kWaves] = {0, 0, 0, 0};
    for (int i = 0; i < N; ++i) {
      const int w = m.owner[i];
      const int st = fill[w];
      for (auto& e : m.nbr[i]) {
        const size_t s = ((size_t)b * kWaves + w) * EW + fill[w]++;
        M.edges[s] = (uint32_t)i | ((uint32_t)e.first << 8);
        M.emask[s] = e.second;
      }
      if (!m.nbr[i].empty()) last_i[w] = i;
      M.seg[(size_t)b * N + i] = ((uint32_t)w << 30) | ((uint32_t)st << 15) | (uint32_t)m.nbr[i].size();
    }
    for (int w = 0; w < kWaves; ++w) {
      const int padded = (fill[w] + 31) / 32 * 32;
      for (int s = fill[w]; s < padded; ++s)  // padding slots: same receiving node, mask 0
        M.edges[((size_t)b * kWaves + w) * EW + s] = (uint32_t)last_i[w] | ((uint32_t)last_i[w] << 8);
      M.npairs[(size_t)b * kWaves + w] = padded / 32;
    }
  }
  M.order.resize(B);
  for (int b = 0; b < B; ++b) M.order[b] = b;
  std::stable_sort(M.order.begin(), M.order.end(), [&](int a, int c) { return mols[a].total > mols[c].total; });
  (void)node_mask;
  return GAUDI_OK;
}

// -------------------------------------------------------------------------------------------------
// graph metadata of the 8-wave kernels: ONE flat slot list per molecule -- live edges sorted by receiving node (then by
// sending node), cut into 16-slot tiles (tile tau -> wave tau & 7, round tau >> 3).  A node's run may straddle two tiles but
// never three (a run that would is moved to the next tile boundary), so its edge -> node sum has at most two partials.
struct Meta8 {
  int S = 16;
  std::vector<int> order, ntiles, ncols;
  std::vector<uint32_t> edges, seg;
  std::vector<float> emask;
  std::vector<uint16_t> soff, sidx;
};

// align: [B][N] flags or nullptr -- a node whose flag is set starts its edge run at a tile boundary (packed launches: the
// first node of every component, so that a molecule's tiles are the ones it has on its own)
static int build_meta8(int B, int N, const float* node_mask, const float* edge_mask, Meta8& M, std::string& err,
                       int min_slots = 0, const uint8_t* align = nullptr) {
  if (N > 255) {
    err = "N > 255 unsupported";
    return GAUDI_E_CAPACITY;
  }
  struct Slot {
    int i, j;
    float m;
    bool live;
  };
  std::vector<std::vector<Slot>> slots(B);
  std::vector<std::vector<std::pair<int, int>>> runs(B);  // per node: (first slot, length)
  M.ncols.assign(B, 1);
  M.ntiles.assign(B, 0);
  std::vector<int> total(B, 0);
  int max_tiles = 1;
  for (int b = 0; b < B; ++b) {
    std::vector<Slot>& sl = slots[b];
    runs[b].assign(N, {0, 0});
    int last = 0;
    bool want_align = false;
    for (int i = 0; i < N; ++i) {
      const bool li = node_mask == nullptr || node_mask[(size_t)b * N + i] != 0.f;
      if (li) last = i;
      std::vector<Slot> run;
      for (int j = 0; j < N; ++j) {
        const float v = edge_mask[((size_t)b * N + i) * N + j];
        const bool lj = node_mask == nullptr || node_mask[(size_t)b * N + j] != 0.f;
        if (v != 0.f && (li || lj)) {  // edges between two masked nodes reach nothing (see build_meta)
          run.push_back({i, j, v, true});
          last = std::max(last, std::max(i, j));
        }
      }
      const int L = (int)run.size();
      if (align != nullptr && align[(size_t)b * N + i]) want_align = true;  // (kept pending over nodes without a run)
      if (L == 0) {
        runs[b][i] = {(int)sl.size(), 0};
        continue;
      }
      if (L > 32) {
        err = "a node with more than 32 live edges is not supported by the 8-wave kernels";
        return GAUDI_E_CAPACITY;
      }
      if (want_align && !sl.empty()) {  // first run of a component: start a new tile
        const int prev_i = sl.back().i;
        while (sl.size() % 16) sl.push_back({prev_i, prev_i, 0.f, false});
      }
      want_align = false;
      const int o = (int)sl.size() % 16;
      if ((o + L + 15) / 16 > 2) {  // would straddle three tiles: start at the next tile boundary
        const int prev_i = sl.empty() ? i : sl.back().i;
        while (sl.size() % 16) sl.push_back({prev_i, prev_i, 0.f, false});
      }
      runs[b][i] = {(int)sl.size(), L};
      sl.insert(sl.end(), run.begin(), run.end());
      total[b] += L;
    }
    if (!sl.empty()) {
      const int prev_i = sl.back().i;
      while (sl.size() % 16) sl.push_back({prev_i, prev_i, 0.f, false});
    }
    M.ncols[b] = last + 1;
    M.ntiles[b] = (int)sl.size() / 16;
    max_tiles = std::max(max_tiles, M.ntiles[b]);
  }
  M.S = std::max(16 * max_tiles, (min_slots + 15) / 16 * 16);  // min_slots: plan hint (slot capacity of the whole logical batch)
  if (M.S > 0xffff) {
    err = "too many edge slots";
    return GAUDI_E_CAPACITY;
  }
  const int S = M.S;
  M.edges.assign((size_t)B * S, 0);
  M.emask.assign((size_t)B * S, 0.f);
  M.seg.assign((size_t)B * N, 0);
  M.soff.assign((size_t)B * (N + 1), 0);
  M.sidx.assign((size_t)B * S, 0);
  for (int b = 0; b < B; ++b) {
    const std::vector<Slot>& sl = slots[b];
    const int ns = (int)sl.size();
    for (int t0 = 0; t0 < ns; t0 += 16) {
      int c = 0;
      while (c < 16) {  // segments of equal receiving node inside the tile (padding extends the segment before it)
        int e = c;
        while (e + 1 < 16 && sl[t0 + e + 1].i == sl[t0 + c].i) ++e;
        const int i = sl[t0 + c].i;
        const int part = runs[b][i].second > 0 && runs[b][i].first < t0 ? 1 : 0;  // the node's run began in an earlier tile
        for (int k = c; k <= e; ++k) {
          const Slot& q = sl[t0 + k];
          M.edges[(size_t)b * S + t0 + k] = (uint32_t)q.i | ((uint32_t)q.j << 8) | ((uint32_t)c << 16) |
                                            ((uint32_t)(k == e) << 20) | ((uint32_t)part << 21);
          M.emask[(size_t)b * S + t0 + k] = q.m;
        }
        c = e + 1;
      }
    }
    for (int i = 0; i < N; ++i)
      M.seg[(size_t)b * N + i] = ((uint32_t)runs[b][i].first << 16) | (uint32_t)runs[b][i].second;
    // slots by SENDING node, ascending slot (= ascending receiving node): the transposed sums of the reverse pass
    uint16_t* off = &M.soff[(size_t)b * (N + 1)];
    uint16_t* idx = &M.sidx[(size_t)b * S];
    int n_out = 0;
    for (int j = 0; j < N; ++j) {
      off[j] = (uint16_t)n_out;
      for (int k = 0; k < ns; ++k)
        if (sl[k].live && sl[k].j == j) idx[n_out++] = (uint16_t)k;
    }
    off[N] = (uint16_t)n_out;
  }
  M.order.resize(B);
  for (int b = 0; b < B; ++b) M.order[b] = b;
  std::stable_sort(M.order.begin(), M.order.end(), [&](int a, int c) {
    return M.ntiles[a] != M.ntiles[c] ? M.ntiles[a] > M.ntiles[c] : total[a] > total[c];
  });
  return GAUDI_OK;
}

// -------------------------------------------------------------------------------------------------
// kernel table: the instantiations live in kern_*.hip (compiled in parallel), each exporting a lookup
typedef void (*kernel_fn)(const KParams);
#ifdef GAUDI_STAMP_STUBS  // diagnostic build: only the two production kernels are linked
#define GAUDI_KERNEL_TUS(X) X(edm_192) X(fused_192_208)
#else
#define GAUDI_KERNEL_TUS(X)                                                                            \
  X(edm_small) X(edm_192) X(edm_208) X(edm_256) X(pred_small) X(pred_192) X(pred_208) X(pred_256)      \
  X(fused_tiny) X(fused_128_128) X(fused_192_192) X(fused_192_208) X(fused_208_208) X(fused_256_256)
#endif
#define X(name) kernel_fn gaudi_kern_##name(int hpe, int hpp);
GAUDI_KERNEL_TUS(X)
#undef X

static kernel_fn pick_kernel(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern_##name(hpe, hpp);
  GAUDI_KERNEL_TUS(X)
#undef X
  return f;
}

// the 4-wave kernels with node buffers in global memory (kerng_*.hip): molecules beyond the LDS limit
#ifdef GAUDI_STAMP_STUBS
static kernel_fn pick_kernel_g(int, int) { return nullptr; }
static bool have_kernels_g(int, int) { return false; }
#else
kernel_fn gaudi_kerng_edm(int hpe, int hpp);
kernel_fn gaudi_kerng_pred(int hpe, int hpp);
// there is no fused (EDM + predictor) V4G instantiation: at the default widths it sits on the register cliff (512 registers,
// 1120 spilled scalars) and faulted; a guided step of a large molecule is two launches, EDM-only then predictor-only
static kernel_fn pick_kernel_g(int hpe, int hpp) {
  kernel_fn f = gaudi_kerng_edm(hpe, hpp);
  if (!f) f = gaudi_kerng_pred(hpe, hpp);
  return f;
}
static bool have_kernels_g(int hpe, int hpp) {
  return (!hpe || pick_kernel_g(hpe, 0)) && (!hpp || pick_kernel_g(0, hpp));
}
#endif

// the 4-wave kernels of sin_embedding denoisers (kernse_*.hip: 24 edge features instead of 2, edm_device.h); the predictor-only
// launches of such a handle (no sin_embedding there) take the ordinary kernels
#ifdef GAUDI_STAMP_STUBS
static kernel_fn pick_kernel_se(int, int, bool) { return nullptr; }
#else
kernel_fn gaudi_kernse_edm(int hpe, int hpp, int gn);
kernel_fn gaudi_kernse_edm_more(int hpe, int hpp, int gn);
kernel_fn gaudi_kernse_fused(int hpe, int hpp, int gn);
static kernel_fn pick_kernel_se(int hpe, int hpp, bool gn) {
  kernel_fn f = gaudi_kernse_edm(hpe, hpp, gn);
  if (!f) f = gaudi_kernse_edm_more(hpe, hpp, gn);
  if (!f) f = gaudi_kernse_fused(hpe, hpp, gn);
  return f;
}
#endif

// the 8-wave instantiations (kern8_*.hip)
#ifdef GAUDI_STAMP_STUBS
#define GAUDI_KERNEL8_TUS(X) X(edm_192) X(fused_192_208)
#else
#define GAUDI_KERNEL8_TUS(X)                                                                           \
  X(edm_small) X(edm_192) X(edm_208) X(edm_256) X(pred_small) X(pred_192) X(pred_208) X(pred_256)      \
  X(fused_tiny) X(fused_128_128) X(fused_192_192) X(fused_192_208) X(fused_208_208) X(fused_256_256)
#endif
#define X(name) kernel_fn gaudi_kern8_##name(int hpe, int hpp);
GAUDI_KERNEL8_TUS(X)
#undef X
static kernel_fn pick_kernel8(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8_##name(hpe, hpp);
  GAUDI_KERNEL8_TUS(X)
#undef X
  return f;
}

// ... and their split-operand (fp16 pairs) versions (kern8s_*.hip); a size without one runs on the fp32-MFMA kernel
#ifdef GAUDI_STAMP_STUBS
#define GAUDI_KERNEL8S_TUS(X) X(edm_192) X(fused_192_208)
#else
#define GAUDI_KERNEL8S_TUS(X) X(edm_small) X(edm_192) X(pred_small) X(pred_208) X(fused_tiny) X(fused_192_208)
#endif
#define X(name) kernel_fn gaudi_kern8s_##name(int hpe, int hpp);
GAUDI_KERNEL8S_TUS(X)
#undef X
static kernel_fn pick_kernel8s(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8s_##name(hpe, hpp);
  GAUDI_KERNEL8S_TUS(X)
#undef X
  return f;
}

// ... the same kernel with FR set (kern8s2_*.hip: sampler_kernel.h, V8T): taken for more than 16 node slots where it exists
#ifdef GAUDI_STAMP_STUBS
#define GAUDI_KERNEL8S2_TUS(X)
#else
#define GAUDI_KERNEL8S2_TUS(X) X(edm_192) X(fused_tiny) X(fused_192_208)
#endif
#define X(name) kernel_fn gaudi_kern8s2_##name(int hpe, int hpp);
GAUDI_KERNEL8S2_TUS(X)
#undef X
static kernel_fn pick_kernel8s2(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8s2_##name(hpe, hpp);
  GAUDI_KERNEL8S2_TUS(X)
#undef X
  (void)hpe; (void)hpp;
  return f;
}

// ... and with the half-size ring (kern8h_*.hip): larger molecules
#ifdef GAUDI_STAMP_STUBS
#define GAUDI_KERNEL8H_TUS(X) X(fused_192_208)
#else
#define GAUDI_KERNEL8H_TUS(X) X(edm_192) X(fused_tiny) X(fused_192_208)
#endif
#define X(name) kernel_fn gaudi_kern8h_##name(int hpe, int hpp);
GAUDI_KERNEL8H_TUS(X)
#undef X
static kernel_fn pick_kernel8h(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8h_##name(hpe, hpp);
  GAUDI_KERNEL8H_TUS(X)
#undef X
  return f;
}
// ... and the kernels whose predictor runs several rounds of edge tiles (kern8m_*.hip: graphs of more than 128 slots)
#if defined(GAUDI_STAMP_STUBS) && !defined(GAUDI_STAMP_M)
static kernel_fn pick_kernel8m(int, int, int) { return nullptr; }
#else
#ifdef GAUDI_STAMP_STUBS  // tools/build_stamped.sh m: the fused MR half-ring kernel (what wide groups run) with phase stamps
#define GAUDI_KERNEL8M_TUS(X) X(fused_192_208_h)
#else
#define GAUDI_KERNEL8M_TUS(X) \
  X(fused_192_208_s) X(fused_192_208_h) X(fused_192_208_f) X(pred_208_s) X(pred_208_h) X(pred_208_f) X(fused_tiny) X(pred_small)
#endif
#define X(name) kernel_fn gaudi_kern8m_##name(int hpe, int hpp, int mode);
GAUDI_KERNEL8M_TUS(X)
#undef X
static kernel_fn pick_kernel8m(int hpe, int hpp, int mode) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8m_##name(hpe, hpp, mode);
  GAUDI_KERNEL8M_TUS(X)
#undef X
  return f;
}
#endif
// ... and the 8-wave kernels with the node buffers in global memory (kern8g_*.hip: V8G)
#if defined(GAUDI_STAMP_STUBS) && !defined(GAUDI_STAMP_G)
static kernel_fn pick_kernel8g(int, int) { return nullptr; }
#else
#ifdef GAUDI_STAMP_STUBS  // tools/build_stamped.sh g: the fused V8G kernel with phase stamps
#define GAUDI_KERNEL8G_TUS(X) X(fused_192_208)
#else
#define GAUDI_KERNEL8G_TUS(X) X(fused_192_208) X(edm_192) X(pred_208) X(fused_tiny) X(edm_small) X(pred_small)
#endif
#define X(name) kernel_fn gaudi_kern8g_##name(int hpe, int hpp);
GAUDI_KERNEL8G_TUS(X)
#undef X
static kernel_fn pick_kernel8g(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8g_##name(hpe, hpp);
  GAUDI_KERNEL8G_TUS(X)
#undef X
  return f;
}
#endif
// ... of which P and Q stay in LDS (kern8gp_*.hip, round 6)
#ifdef GAUDI_STAMP_STUBS
static kernel_fn pick_kernel8gp(int, int) { return nullptr; }
#else
#define GAUDI_KERNEL8GP_TUS(X) X(fused_192_208) X(edm_192) X(pred_208) X(tiny)
#define X(name) kernel_fn gaudi_kern8gp_##name(int hpe, int hpp);
GAUDI_KERNEL8GP_TUS(X)
#undef X
static kernel_fn pick_kernel8gp(int hpe, int hpp) {
  kernel_fn f = nullptr;
#define X(name) \
  if (!f) f = gaudi_kern8gp_##name(hpe, hpp);
  GAUDI_KERNEL8GP_TUS(X)
#undef X
  return f;
}
#endif
// ... and the wide-group kernels on the full ring (kern8mp_fused.hip, round 6)
#ifdef GAUDI_STAMP_STUBS
static kernel_fn pick_kernel8mp(int, int) { return nullptr; }
#else
kernel_fn gaudi_kern8mp_fused(int hpe, int hpp);
static kernel_fn pick_kernel8mp(int hpe, int hpp) { return gaudi_kern8mp_fused(hpe, hpp); }
#endif
// mr: the call holds a graph of more than one round of edge tiles AND runs the predictor
static kernel_fn pick_kernel8_mode(int hpe, int hpp, int mode, bool mr = false) {
  if (mr) return pick_kernel8m(hpe, hpp, mode);
  return mode == 1 ? pick_kernel8s(hpe, hpp) : mode == 2 ? pick_kernel8h(hpe, hpp) : pick_kernel8(hpe, hpp);
}

// smallest instantiated padded hidden size >= H (0 if none)
static int round_hidden(int H) {
  static const int sizes[] = {32, 48, 64, 128, 192, 208, 256};
  for (int s : sizes)
    if (s >= H) return s;
  return 0;
}

// gn: node buffers in global memory (V4G kernels)
// edge slots per wave of the 4-wave kernels for the fully connected graph of N nodes (build_meta: whole receiving nodes are
// dealt to the waves, the longest list rounded up to 32)
static int dense_ew4(int N) { return std::max(32, ((N + kWaves - 1) / kWaves * (N - 1) + 31) / 32 * 32); }

// ef: edge features of the denoiser's first Linears (2, or 24 for sin_embedding: edm_device.h, NetSmem)
static size_t lds_bytes(int hpe, int hpp, int N, int D, int EW, bool gn = false, int ef = 2) {
  size_t net = 0;
  if (hpe)
    net = std::max(net, (size_t)((gn ? 0 : 4 * N * (hpe + 4)) + kWaves * 16 * (hpe + 4) + 8 * N + kWaves * EW * 9 + (6 + ef) * hpe +
                                 (ef > 2 ? kWaves * EW * ef : 0)));
  if (hpp) net = std::max(net, (size_t)((gn ? 0 : 5 * N * (hpp + 4)) + kWaves * 16 * (hpp + 4) + 12 * N + kWaves * EW * 10 + 32 + 10 * hpp));
  return sizeof(float) * (common_floats(N, D, EW) + net);
}
static size_t gnode_floats(int hpe, int hpp, int N) {
  return std::max((size_t)(hpe ? 4 * N * (hpe + 4) : 0), (size_t)(hpp ? 5 * N * (hpp + 4) : 0));
}

// gn: 0 resident, 1 the five node buffers in global memory, 2 of which P / Q in LDS (w8_edm.h: gn_lds_buffers); 3 (round 6: wide
// groups on the full ring) the resident kernels with the PREDICTOR's fifth node buffer in global memory (w8_pred.h: PredSmem, PG)
static size_t lds_floats8_base(int hpe, int hpp, int N, int D, int S, int split, int gn = 0) {
  size_t net = 0;
  const int pbuf = gn == 3 ? 4 : 5;
  if (gn == 3) gn = 0;
  if (hpe) net = std::max(net, (size_t)((gn ? w8::gn_lds_buffers(gn) : 5) * N * (hpe + 4) + w8::edge_ring_floats(hpe, split) + 8 * N + 2 * align4(N) + 96 + S * 9 + 8 * hpe));
  if (hpp) net = std::max(net, (size_t)(w8::edge_ring_floats(hpp, split) + (gn ? w8::gn_lds_buffers(gn) : pbuf) * N * (hpp + 4) + 12 * N + 2 * align4(N) + 96 + S * 10 + 32 + 10 * hpp));
  return common_floats8(N, D, S) + net;
}
static size_t gnode_floats8(int hpe, int hpp, int N) {
  return std::max((size_t)(hpe ? 5 * N * (hpe + 4) : 0), (size_t)(hpp ? 5 * N * (hpp + 4) : 0));
}
// V8G stages up to two node buffers in the idle weight ring before each node GEMM (w8_common.h: stage_rows)
static bool gn8_stage_fits(int hpe, int hpp, int N) {
  for (int hp : {hpe, hpp})
    if (hp && 2 * (size_t)w8::stage_stride(N * (hp + 4)) > w8::edge_ring_floats(hp, 1)) return false;
  return true;
}
// The reverse pass publishes du of every slot pub_ch feature tiles at a time into [b0 | b1 | pubx extra floats]: pick the
// largest pub_ch that fits 160 KiB, then the extra floats that choice needs.  false: the molecule does not fit.
// fp16-pair node GEMMs (every split-operand kernel): the split copy of ONE GEMM input must fit the region the kernels use -- the
// ring's free slot (full ring), the whole ring where it idles across node phases (half ring, gn) -- for N node columns
static bool node_f16_fits(int hp, int N, int split, bool gn) {
  if (!hp || !split || !GAUDI_NODE_F16) return true;
  if (N > (gn ? 48 : 32)) return false;  // column tiles of one pass: three in the gn kernels, two in the resident ones
  const int nct = N <= 16 ? 1 : N <= 32 ? 2 : 3;
  const int ring = w8::edge_ring_floats(hp, split);
  return w8::nh_split_floats(hp, nct) <= (w8::node_ring_idle(hp, split, gn) ? ring : ring / 2);
}
static bool plan_pub8(int hpe, int hpp, int N, int D, int S, int split, int& pubx, int& pub_ch, int gn = 0) {
  pubx = 0;
  pub_ch = 0;
  const int gn_lds = gn;  // (3: the resident kernels' planning with four predictor buffers in LDS)
  if (gn == 3) gn = 0;
  if (!node_f16_fits(hpe, N, split, gn != 0) || !node_f16_fits(hpp, N, split, gn != 0)) return false;
  const long long cap = 160 * 1024 / 4 - 64;  // floats (a little headroom for the runtime's own static LDS)
  const long long base = (long long)lds_floats8_base(hpe, hpp, N, D, S, split, gn_lds);
  if (base > cap) return false;
  if (!hpp) return true;
  const int T = hpp / 16;
  // b0 + b1 double as the head of the publish buffer; with the split edge GEMMs the (then idle) weight ring follows them
  // (V8G: b0 / b1 are in global memory, the buffer is the ring and what follows it)
  const long long own = (gn ? 0LL : 2LL * N * (hpp + 4)) + (split ? w8::edge_ring_floats(hpp, split) : 0);
  pub_ch = w8::pub_chunk_tiles(S, own + (cap - base), T);
  if (pub_ch < 1) return false;
  // prefer the smallest chunk that gives the same number of chunks (less LDS, same barriers)
  const int nch = (T + pub_ch - 1) / pub_ch;
  pub_ch = (T + nch - 1) / nch;
  pubx = (int)std::max(0LL, (long long)S * (16 * pub_ch + 4) - own);
  return true;
}
static size_t lds_bytes8(int hpe, int hpp, int N, int D, int S, int pubx, int split, int gn = 0) {
  return sizeof(float) * (lds_floats8_base(hpe, hpp, N, D, S, split, gn) + (hpp ? pubx : 0));
}

static int launch(gaudi_handle* h, const KParams& P, int hpe, int hpp, long long steps) {
  const bool v8 = h->run_variant == 8;
  const bool se = hpe && h->ecfg.sin_embedding;  // stage_graph keeps such a call on the 4-wave family
  kernel_fn fn = v8   ? (h->run_pg && hpe && hpp ? pick_kernel8mp(hpe, hpp) : h->run_gn8 == 2 ? pick_kernel8gp(hpe, hpp) : h->run_gn8 ? pick_kernel8g(hpe, hpp) : pick_kernel8_mode(hpe, hpp, h->run_split, h->run_mr && hpp))
                 : se ? pick_kernel_se(hpe, hpp, h->run_gn)
                      : h->run_gn ? pick_kernel_g(hpe, hpp) : pick_kernel(hpe, hpp);
  // two column tiles per node GEMM on the resident full-ring kernel: its FR instantiation (same arithmetic, same results)
  if (v8 && !h->run_gn8 && h->run_split == 1 && !(h->run_mr && hpp) && P.N > 16 && !getenv("GAUDI_NO_FR"))
    if (kernel_fn f2 = pick_kernel8s2(hpe, hpp)) fn = f2;
  if (!fn)
    return fail(h, GAUDI_E_INVALID,
                "no kernel instantiated for padded hidden sizes (" + std::to_string(hpe) + "," + std::to_string(hpp) + ")" +
                    (v8 ? " in the 8-wave family" : se ? " among the sin_embedding kernels" : ""));
  size_t lds = v8 ? lds_bytes8(hpe, hpp, P.N, 3 + P.F, P.EW, P.pubx, h->run_split, h->run_pg && hpe && hpp ? 3 : h->run_gn8)
                  : lds_bytes(hpe, hpp, P.N, 3 + P.F, P.EW, h->run_gn, se ? 24 : 2);
  if (v8 && P.hk_off) lds = sizeof(float) * ((size_t)P.hk_off + (size_t)h->run_hk);  // the kept split copy of h sits behind the FUSED plan
  if (lds > 160 * 1024)
    return fail(h, GAUDI_E_CAPACITY, "molecule needs " + std::to_string(lds) + " B of LDS (>160 KiB): N too large");
  {
    // hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function, not to a handle: one process-wide record per
    // (device, kernel), only ever raised, so that no handle launches with more dynamic LDS than the attribute allows
    static std::mutex attr_mu;
    static std::map<std::pair<int, const void*>, int> granted_by_fn;
    std::lock_guard<std::mutex> lock(attr_mu);
    int& granted = granted_by_fn[{h->device, (const void*)fn}];
    if ((int)lds > granted) {
      HIPCHECK(h, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      granted = (int)lds;
    }
  }
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  if (h->prof) HIPCHECK(h, h->prof_log.begin(h->stream, ev));
#ifdef GAUDI_STAMPS
  KParams PS = P;
  HIPCHECK(h, h->d_stamps.reserve(sizeof(unsigned long long) * 32));
  HIPCHECK(h, hipMemsetAsync(h->d_stamps.p, 0, sizeof(unsigned long long) * 32, h->stream));
  PS.stamps = h->d_stamps.as<unsigned long long>();
  hipLaunchKernelGGL(fn, dim3(P.B), dim3(v8 ? w8::kThreads : kThreads), lds, h->stream, PS);
  {
    unsigned long long tmp[32];
    HIPCHECK(h, hipMemcpyAsync(tmp, h->d_stamps.p, sizeof(tmp), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 32; ++i) h->stamp_acc[i] += tmp[i];
    if (getenv("GAUDI_PRINT_STAMPS")) {
      static const char* nm[] = {"node_gemm", "edge_gemm", "edge_epilogue", "barrier", "misc", "bwd_node", "bwd_stash_reload",
                                 "bwd_colsum_tail", "pub_wait", "stash", "pub_write", "colsum_loop", "colsum_barrier", "b_dcp",
                                 "b_gemm_de", "b_dv", "b_gemm_dt1", "b_du", "stage_vectors", "geo", "edm_embed_head", "noise_update",
                                 "bwd_reload_loads", "bwd_vec_commit", "pred_readout", "bwd_prologue", "x2", "x3"};
      unsigned long long tot = 0;
      for (int i = 0; i < ST_N; ++i) tot += h->stamp_acc[i];
      fprintf(stderr, "[stamps] cumulative shares (block 0, wave 0):");
      for (int i = 0; i < ST_N; ++i) fprintf(stderr, " %s=%.1f%%", nm[i], 100.0 * h->stamp_acc[i] / (tot ? tot : 1));
      fprintf(stderr, " total=%llu ticks\n", tot);
      if (tmp[31])
        fprintf(stderr, "[stamps] shader clock during this launch: %.0f MHz (%llu cycles / %llu ticks of 100 MHz)\n",
                100.0 * (double)tmp[30] / (double)tmp[31], tmp[30], tmp[31]);
    }
  }
#else
  if (h->prof) {  // profiled launches also leave their clock counters (sampler_kernel.h: KParams::clock_out)
    KParams PC = P;
    HIPCHECK(h, h->d_clock.reserve(sizeof(unsigned long long) * 4));
    PC.clock_out = h->d_clock.as<unsigned long long>();
    hipLaunchKernelGGL(fn, dim3(P.B), dim3(v8 ? w8::kThreads : kThreads), lds, h->stream, PC);
  } else {
    hipLaunchKernelGGL(fn, dim3(P.B), dim3(v8 ? w8::kThreads : kThreads), lds, h->stream, P);
  }
#endif
  HIPCHECK(h, hipGetLastError());
  if (h->prof) {
    HIPCHECK(h, h->prof_log.end(h->stream, ev));
    h->prof_steps += steps;
  }
  return GAUDI_OK;
}

// Packed launches: groups of molecules, each group one graph of N node slots
struct Pack {
  int G = 0;
  int NG = 0;                         // node slots per group (= N, or more: wide groups)
  std::vector<float> umask, uemask;   // [G][NG], [G][NG][NG]: the union graphs
  std::vector<uint8_t> align;         // [G][NG]: first slot of every component (tile alignment in build_meta8)
  std::vector<int32_t> rowmap;        // [G][NG]: slot -> molecule * N + node | component << 28, or -1
  std::vector<int32_t> compmol;       // [G][kMaxComp]
  std::vector<int32_t> ncomp;         // [G]
};
// NG node slots and TG edge tiles per group (N and 8: the classic packing; more: wide groups, whose edge phases run in rounds)
// nodes a molecule needs slots for in a packed launch: live ones and any node that touches a live edge, in their own order
static std::vector<std::vector<int>> used_nodes(int B, int N, const float* node_mask, const float* edge_mask) {
  std::vector<std::vector<int>> used(B);
  for (int b = 0; b < B; ++b) {
    std::vector<char> keep(N, 0);
    for (int i = 0; i < N; ++i) {
      const bool li = node_mask[(size_t)b * N + i] != 0.f;
      if (li) keep[i] = 1;
      for (int j = 0; j < N; ++j) {
        const bool lj = node_mask[(size_t)b * N + j] != 0.f;
        if (edge_mask[((size_t)b * N + i) * N + j] != 0.f && (li || lj)) keep[i] = keep[j] = 1;
      }
    }
    for (int i = 0; i < N; ++i)
      if (keep[i]) used[b].push_back(i);
  }
  return used;
}
// narrow: NG may be SMALLER than N (every molecule must then fit NG slots on its own: the caller checks) -- rows of the global
// arrays keep the stride N.  maxcomp: molecules per group (1: every molecule alone, its nodes compacted to the front slots).
// molmap: molecule -> index the noise is keyed with (a bucket of a larger request), or nullptr.
// max_multi: groups that may hold more than one molecule (a MIXED wide launch: pairs first, the rest alone in the same kernel).
static void pack_groups(int B, int N, const float* node_mask, const float* edge_mask, const Meta8& M, Pack& pk, int NG = 0,
                        int TG = w8::kWaves, bool narrow = false, int maxcomp = kMaxComp, const int32_t* molmap = nullptr,
                        int max_multi = INT_MAX) {
  if (NG < N && !narrow) NG = N;
  const std::vector<std::vector<int>> used = used_nodes(B, N, node_mask, edge_mask);
  struct Group {
    std::vector<int> mols;
    int nodes = 0, tiles = 0;
  };
  std::vector<Group> groups;
  int multi = 0;
  for (int b : M.order) {  // heaviest first
    const int nn = (int)used[b].size(), nt = M.ntiles[b];
    Group* fit = nullptr;
    for (Group& g : groups)
      if ((int)g.mols.size() < maxcomp && g.nodes + nn <= NG && g.tiles + nt <= TG && (g.mols.size() > 1 || multi < max_multi)) {
        fit = &g;
        if (g.mols.size() == 1) ++multi;
        break;
      }
    if (!fit) {
      groups.emplace_back();
      fit = &groups.back();
    }
    fit->mols.push_back(b);
    fit->nodes += nn;
    fit->tiles += nt;
  }
  const int G = (int)groups.size();
  pk.G = G;
  pk.NG = NG;
  pk.umask.assign((size_t)G * NG, 0.f);
  pk.uemask.assign((size_t)G * NG * NG, 0.f);
  pk.align.assign((size_t)G * NG, 0);
  pk.rowmap.assign((size_t)G * NG, -1);
  pk.compmol.assign((size_t)G * kMaxComp, 0);
  pk.ncomp.assign(G, 0);
  std::vector<int> slot_of(N);
  for (int g = 0; g < G; ++g) {
    int base = 0;
    for (size_t k = 0; k < groups[g].mols.size(); ++k) {
      const int b = groups[g].mols[k];
      pk.compmol[(size_t)g * kMaxComp + k] = molmap ? molmap[b] : b;
      std::fill(slot_of.begin(), slot_of.end(), -1);
      for (size_t r = 0; r < used[b].size(); ++r) slot_of[used[b][r]] = base + (int)r;
      if (!used[b].empty()) pk.align[(size_t)g * NG + base] = 1;
      for (int i : used[b]) {
        const int si = slot_of[i];
        pk.umask[(size_t)g * NG + si] = node_mask[(size_t)b * N + i];
        pk.rowmap[(size_t)g * NG + si] = (b * N + i) | ((int32_t)k << 28);
        for (int j : used[b])
          pk.uemask[((size_t)g * NG + si) * NG + slot_of[j]] = edge_mask[((size_t)b * N + i) * N + j];
      }
      base += (int)used[b].size();
    }
    pk.ncomp[g] = (int)groups[g].mols.size();
  }
}

// upload masks + metadata, fill the graph part of KParams
// -> GAUDI_OK, or a positive value = "run this call on the 4-wave kernels" (graph outside the 8-wave kernels' limits)
static int stage_graph8(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, KParams& P, int hpe,
                        int hpp) {
  if (std::max(h->plan_force_waves, h->call_force_waves) == 4) return 1;  // the whole logical batch runs on 4 waves
  Meta8 M;
  std::string err;
  int rc = build_meta8(B, N, node_mask, edge_mask, M, err, std::max(h->plan_min_slots, h->call_min_slots));
  if (rc == GAUDI_E_CAPACITY) return 1;
  if (rc) return fail(h, rc, err);
  // more than one round of eight edge tiles (a graph of more than 128 live-edge slots, e.g. a fully connected molecule of
  // 12+ nodes): the 8-wave kernels run the rounds one after the other (GAUDI_PRED_ROUNDS=0: the round-2 behaviour, such calls
  // go to the 4-wave kernels); the LDS plan below decides whether the larger per-slot arrays and publish buffer still fit
  const bool mr = hpp && (M.S > 16 * w8::kWaves || h->force_mr);
  if (mr && !h->pred_rounds) return 1;
  // ---- packing: small molecules share a workgroup as the components of one graph (sampling calls only).  First-fit in
  // the heaviest-first order; a group holds at most kMaxComp molecules, N node slots and 8 edge tiles (one round on 8
  // waves).  A molecule keeps its own tiles (component starts are tile-aligned), its nodes keep their relative order, its
  // noise is keyed by its own sample and node indices, and every per-molecule reduction runs per component in the order
  // the molecule's own workgroup would use: the result does not depend on the packing, bit for bit.
  if (!pick_kernel8_mode(hpe, hpp, 0, mr) && !pick_kernel8_mode(hpe, hpp, 1, mr) && !pick_kernel8_mode(hpe, hpp, 2, mr)) return 1;
  const int Dz = 3 + (hpe ? h->ecfg.in_node_nf : h->pcfg.in_nf);
  int pubx = 0, pub_ch = 0;
  // the arithmetic of the edge GEMMs for S edge slots on NS node slots: split operands when the kernel exists and its larger weight
  // ring fits (1 = full ring, 2 = half ring); else fp32 MFMAs (0); else -1 = this call runs on 4 waves
  // the split-operand kernels run their node GEMMs on fp16 pairs: a weight set whose images were refused (NodeScale) keeps the
  // fp32-instruction kernels
  const bool node_f16_ok = !GAUDI_NODE_F16 || ((!hpe || h->edm_hinv > 0.f) && (!hpp || h->pred_hinv > 0.f));
  auto plan_for = [&](int NS, int S, bool mrk) -> int {
    if (h->split && (!hpe || h->edm_ws_bytes) && (!hpp || h->pred_ws_bytes) && node_f16_ok)
      for (int mode = 1; mode <= 2; ++mode)
        if (pick_kernel8_mode(hpe, hpp, mode, mrk) && plan_pub8(hpe, hpp, NS, Dz, S, mode, pubx, pub_ch)) return mode;
    return pick_kernel8_mode(hpe, hpp, 0, mrk) && plan_pub8(hpe, hpp, NS, Dz, S, 0, pubx, pub_ch) ? 0 : -1;
  };
  int mode_u = plan_for(N, M.S, mr);
  Pack pk;
  const int B0 = B;
  const float* nm_used = node_mask;
  int n_slots = N, mode_run = mode_u;
  bool mr_run = mr;
  bool narrow_taken = false, use_pack = false, pg_run = false;
  // Round 6: a bucket of molecules that fit the resident kernels on their own (gaudi_sample: per-molecule kernel family) while the
  // call's padded N does not -- packed groups of call_narrow (< N) node slots, one round of eight edge tiles.
  if (h->call_narrow > 0 && h->call_narrow < N && h->pack_now && (int64_t)B * N < (1 << 28)) {
    pack_groups(B, N, node_mask, edge_mask, M, pk, h->call_narrow, w8::kWaves, true, kMaxComp, h->call_molmap);
    Meta8 M2;
    rc = build_meta8(pk.G, h->call_narrow, pk.umask.data(), pk.uemask.data(), M2, err, 0, pk.align.data());
    if (rc != GAUDI_OK) return fail(h, rc, "per-molecule kernel family: " + err);
    const bool mr2 = hpp && M2.S > 16 * w8::kWaves;
    const int mode2 = plan_for(h->call_narrow, M2.S, mr2);
    if (mode2 < 1 || mr2) return fail(h, GAUDI_E_CAPACITY, "per-molecule kernel family: the resident plan of the small bucket does not fit");
    M = std::move(M2);
    B = pk.G;
    nm_used = pk.umask.data();
    n_slots = h->call_narrow;
    mode_run = mode_u = mode2;
    mr_run = mr2;
    narrow_taken = use_pack = true;
  }
  // V8G (round 4): a molecule whose node buffers do not fit LDS beside the ring runs on the 8-wave kernels with those five
  // buffers in a per-workgroup global scratch (split edge GEMMs, full ring, several rounds of edge tiles) -- before round 4
  // such calls fell to the 4-wave V4G kernels (fp32 matrix instructions, two launches per guided step)
  // Round 6: of the five, P and Q stay in LDS where that plan fits (kern8gp_*.hip; +3.6 % on 40-node molecules) -- a function of
  // the widths, N and the edge slots the plan is made with (the whole batch's: call_min_slots / plan_min_slots), like the rest
  int gn8 = 0;
  if (!narrow_taken && (mode_u < 0 || h->force_gn8) && h->gn8 && h->split && (!hpe || h->edm_ws_bytes) && (!hpp || h->pred_ws_bytes) && node_f16_ok &&
      (GAUDI_NODE_F16 || gn8_stage_fits(hpe, hpp, N))) {
    if (h->gn8_pq && GAUDI_NODE_F16 && pick_kernel8gp(hpe, hpp) && plan_pub8(hpe, hpp, N, Dz, M.S, 1, pubx, pub_ch, 2)) gn8 = 2;
    else if (pick_kernel8g(hpe, hpp) && plan_pub8(hpe, hpp, N, Dz, M.S, 1, pubx, pub_ch, 1)) gn8 = 1;
  }
  if (gn8) {
    mode_u = 1;
  } else if (h->force_gn8 && mode_u >= 0) {
    plan_for(N, M.S, mr);  // (restore pubx / pub_ch of the resident plan)
  }
  if (mode_u < 0) return 1;
  if (!narrow_taken) mode_run = mode_u;
  // V8G sampling calls (round 6): every molecule alone in its workgroup as before, but its nodes COMPACTED to the front slots
  // (a hetero molecule's rings and orientation nodes are two blocks of the padded index range: n rings occupy columns up to
  // N / 2 + n) -- fewer node-GEMM column tiles; the packed form also carries the molecule's index in the request (call_molmap).
  if (gn8 && h->pack_now && h->pack && h->gn8_pack && (int64_t)B * N < (1 << 28)) {
    pack_groups(B, N, node_mask, edge_mask, M, pk, N, 1 << 20, false, 1, h->call_molmap);
    Meta8 M2;
    int pubx2 = 0, pub_ch2 = 0;
    if (build_meta8(pk.G, N, pk.umask.data(), pk.uemask.data(), M2, err, M.S, pk.align.data()) == GAUDI_OK && M2.S == M.S &&
        plan_pub8(hpe, hpp, N, Dz, M2.S, 1, pubx2, pub_ch2, gn8)) {
      M = std::move(M2);
      B = pk.G;
      nm_used = pk.umask.data();
      pubx = pubx2;
      pub_ch = pub_ch2;
      use_pack = true;
    } else {
      if (h->call_molmap) return fail(h, GAUDI_E_CAPACITY, "per-molecule kernel family: the packed V8G plan does not fit");
      pk = Pack();
    }
  }
  // (a row of the map holds molecule * N + node in 28 bits)
  if (!gn8 && !narrow_taken && h->pack_now && h->pack && B > 1 && (int64_t)B * N < (1 << 28)) {
    // Candidate group shapes, widest first.  WIDE groups (opt-in: GAUDI_PAIRS=1 for batches of at least two molecules per CU, 2
    // always): up to 2 N node slots and two rounds of eight edge tiles -- e.g. two 11-ring cata molecules, or three to four small
    // hetero ones, per workgroup.  Every node-level matrix is then streamed from L2 once for all of them and the per-GEMM fixed
    // costs are shared; the edge phases run their rounds one after the other on the half ring (the predictor on the MR
    // kernels).  Bit-identical to the narrow launch.  Then the classic shape: N node slots, one round.
    struct Cand {
      int NG, TG;
    };
    std::vector<Cand> cands;
    // (pairs == 1, the default: when the batch holds at least two molecules per CU AND the classic shape packs next to nothing --
    // cata molecules fill their N slots; a hetero batch whose small molecules already share workgroups gains nothing from the
    // second round: C4 -0.9 %, C3 at 1 024 molecules +2.6 %, C2 at 1 024 +6.7 % with the fp16-pair node GEMMs, profiles/r05c_*)
    // Round 6: "pays" is decided by ROUNDS of workgroups on the chip -- a workgroup of two molecules takes 1.81 x one molecule's time
    // (111.5 against 2 x 30.8 ms per launch, profiles/r06h_wide_pairs*), so pairs win where they save enough rounds: 257-512 and
    // 769-1024 molecules on 256 CUs (2 -> 1.8 and 4 -> 3.6 round-times), not 513-768 (3 -> 3.6).  Same bits either way.
    // MIXED launches: with p two-molecule groups and B - 2 p single ones in the same wide kernel (heaviest first: the pairs), the
    // makespan on the chip's CUs is what a list schedule gives -- wide_plan tries p = every multiple of the CU count and "all pairs",
    // and takes the wide kernel where its best beats ceil(B / CUs) rounds of the one-molecule kernel: 640 molecules run as 256 pairs +
    // 128 single ones (2.84 round-times against 3), 1 024 as 512 pairs, 256 one per workgroup.
    bool want_wide = h->pairs == 2;
    int max_multi = INT_MAX;
    if (h->pairs == 1 && B > h->num_cus) {
      const int cus = std::max(1, h->num_cus);
      const double solo = (double)((B + cus - 1) / cus);
      auto makespan = [&](int p) {  // list schedule of p jobs of kPair and B - 2 p jobs of kAlone on `cus` machines, longest first
        // (measured at the default widths, profiles/r06k_wide_group_rule.txt: a pair 2 228 ms, a molecule alone in the wide kernel 1 325 ms,
        // the one-molecule kernel 1 236 ms per 1 000 steps)
        const double kPair = 1.81, kAlone = 1.07;
        std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
        for (int c = 0; c < cus; ++c) free_at.push(0.0);
        double end = 0.0;
        auto run = [&](int n, double cost) {
          for (int k = 0; k < n; ++k) {
            const double t = free_at.top() + cost;
            free_at.pop();
            free_at.push(t);
            end = std::max(end, t);
          }
        };
        run(p, kPair);
        run(B - 2 * p, kAlone);
        return end;
      };
      double best = solo * 0.995;  // (the one-molecule kernel unless the wide one is clearly ahead)
      int best_p = -1;
      for (int p = cus; p <= B / 2 + cus - 1; p += cus) {
        const int pp = std::min(p, B / 2);
        const double m = makespan(pp);
        if (m < best) { best = m; best_p = pp; }
      }
      if (h->pairs_cap >= 0) best_p = std::min(h->pairs_cap, B / 2);
      if (best_p >= 0) {
        pack_groups(B, N, node_mask, edge_mask, M, pk, N, w8::kWaves);
        want_wide = (long long)pk.G * 10 >= (long long)B * 9;
        max_multi = best_p;
      }
    }
    if (want_wide && h->pairs_cap >= 0) max_multi = h->pairs_cap;  // (experiments)
    if (want_wide)
      for (int ng = std::min(2 * N, 32); ng > N; --ng) {
        const long long cap = 160 * 1024 / 4 - 64;
        if ((long long)lds_floats8_base(hpe, hpp, ng, Dz, 16 * (w8::kWaves + 1), h->split ? 2 : 0) > cap) continue;  // cannot fit whatever the slots
        cands.push_back({ng, 2 * w8::kWaves});
      }
    cands.push_back({N, w8::kWaves});
    bool taken = false;
    for (const Cand& cd : cands) {
      const bool wide = cd.NG > N;
      pack_groups(B, N, node_mask, edge_mask, M, pk, cd.NG, cd.TG, false, kMaxComp, nullptr, wide ? max_multi : INT_MAX);
      if (pk.G >= B && !(wide && h->pairs_cap == 0)) continue;  // (GAUDI_PAIRS_CAP=0: every molecule alone in the wide kernel -- diagnostic)
      Meta8 M2;
      rc = build_meta8(pk.G, cd.NG, pk.umask.data(), pk.uemask.data(), M2, err, wide ? 0 : M.S, pk.align.data());
      if (rc != GAUDI_OK) continue;
      const bool mr2 = hpp && M2.S > 16 * w8::kWaves;
      if (!wide && M2.S > std::max(M.S, 16 * w8::kWaves)) continue;  // (a classic group holds <= 8 tiles)
      if (mr2 && !mr && !h->pred_rounds) continue;
      // a packed launch has more edge slots per workgroup; it must keep the ARITHMETIC the unpacked plan has -- split operands
      // (full or half ring: the same sums in the same order) or fp32 instructions -- because the plan of a sharded batch is the
      // same on every rank (gaudi_set_plan_hint) and packing must not move a rank off it
      int mode2 = plan_for(cd.NG, M2.S, mr2);
      if (mode2 < 0 || (mode2 != 0) != (mode_u != 0)) continue;
      // Round 6: a wide group that fits the HALF ring only (two cata-11 molecules at the default widths: five predictor buffers of 22
      // node slots + the 52 KiB ring are 171 KB) runs on the FULL ring with the predictor's fifth node buffer in the workgroup's global
      // scratch (kern8mp_fused.hip) -- the same sums in the same order, one trip per K chunk instead of two
      pg_run = false;
      if (wide && mode2 == 2 && mr2 && h->wide_full && hpe && hpp && pick_kernel8mp(hpe, hpp)) {
        if (plan_pub8(hpe, hpp, cd.NG, Dz, M2.S, 1, pubx, pub_ch, 3)) {
          mode2 = 1;
          pg_run = true;
        } else {
          plan_for(cd.NG, M2.S, mr2);  // (restore pubx / pub_ch of the half-ring plan)
        }
      }
      M = std::move(M2);
      B = pk.G;
      nm_used = pk.umask.data();
      n_slots = cd.NG;
      mode_run = mode2;
      mr_run = mr2;
      taken = use_pack = true;
      break;
    }
    if (!taken) {
      pg_run = false;
      plan_for(N, M.S, mr);  // keep the unpacked plan (pubx / pub_ch)
    }
  }
  const bool packed = use_pack;
  h->run_split = mode_run;
  P.pubx = pubx;
  P.pub_ch = pub_ch;
  if (getenv("GAUDI_DEBUG_PLAN"))
    fprintf(stderr, "[plan] molecules=%d workgroups=%d N=%d node slots=%d S=%d split=%d mr=%d pub_ch=%d pubx=%d lds=%zu\n", B0, B, N, n_slots,
            M.S, h->run_split, gn8 ? 2 : pg_run ? 3 : (int)mr_run, pub_ch, pubx, lds_bytes8(hpe, hpp, n_slots, Dz, M.S, pubx, h->run_split, pg_run ? 3 : gn8));
  auto up = [&](DevBuf& d, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = d.reserve(bytes);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, h->stream);
  };
  HIPCHECK(h, up(h->d_ncols, M.ncols.data(), sizeof(int) * B));
  HIPCHECK(h, up(h->d_mask, nm_used, sizeof(float) * B * n_slots));
  if (packed) {
    HIPCHECK(h, up(h->d_rowmap, pk.rowmap.data(), sizeof(int32_t) * pk.rowmap.size()));
    HIPCHECK(h, up(h->d_compmol, pk.compmol.data(), sizeof(int32_t) * pk.compmol.size()));
    HIPCHECK(h, up(h->d_ncomp, pk.ncomp.data(), sizeof(int32_t) * pk.ncomp.size()));
  }
  HIPCHECK(h, up(h->d_order, M.order.data(), sizeof(int) * B));
  HIPCHECK(h, up(h->d_edges, M.edges.data(), sizeof(uint32_t) * M.edges.size()));
  HIPCHECK(h, up(h->d_emask, M.emask.data(), sizeof(float) * M.emask.size()));
  HIPCHECK(h, up(h->d_npairs, M.ntiles.data(), sizeof(int) * B));
  HIPCHECK(h, up(h->d_seg, M.seg.data(), sizeof(uint32_t) * M.seg.size()));
  HIPCHECK(h, up(h->d_soff, M.soff.data(), sizeof(uint16_t) * M.soff.size()));
  HIPCHECK(h, up(h->d_sidx, M.sidx.data(), sizeof(uint16_t) * M.sidx.size()));
  HIPCHECK(h, hipStreamSynchronize(h->stream));  // M goes out of scope
  P.B = B;
  P.N = n_slots;
  P.NR = N;
  P.EW = M.S;
  P.node_mask = h->d_mask.as<float>();
  P.order = h->d_order.as<int>();
  P.edges = h->d_edges.as<uint32_t>();
  P.emask = h->d_emask.as<float>();
  P.npairs = h->d_npairs.as<int>();
  P.seginfo = h->d_seg.as<uint32_t>();
  P.ncols = h->d_ncols.as<int>();
  P.soff = h->d_soff.as<uint16_t>();
  P.sidx = h->d_sidx.as<uint16_t>();
  P.rowmap = packed ? h->d_rowmap.as<int32_t>() : nullptr;
  P.compmol = packed ? h->d_compmol.as<int32_t>() : nullptr;
  P.ncomp = packed ? h->d_ncomp.as<int32_t>() : nullptr;
  h->run_groups = B;
  h->run_nslots = n_slots;
  h->run_mr = mr_run;
  h->run_gn8 = gn8;
  h->run_pg = pg_run;
  // Kept split copy of h (w8_nodes_f16.h: node_ctx_keep): behind everything the plan of BOTH networks needs, when 160 KiB leave the
  // room -- C2 / C3 do (46 KB free), 20-22 node slots do not.  Same results either way (the copy is a function of h alone).
  P.hk_off = 0;
  h->run_hk = 0;
  if (h->keep_h && GAUDI_NODE_F16 && mode_run >= 1 && !gn8 && !pg_run) {
    const size_t plan = lds_bytes8(hpe, hpp, n_slots, Dz, M.S, pubx, mode_run, false) / sizeof(float);
    const size_t at = (plan + 3) & ~(size_t)3;
    const size_t need = (size_t)w8::nh_keep_floats(std::max(hpe, hpp), n_slots);
    if ((at + need) * sizeof(float) + 1024 <= 160 * 1024) {
      P.hk_off = (int)at;
      h->run_hk = (int)need;
    }
  }
  if (pg_run) {  // one [node slots][hpp + 4] buffer per workgroup
    const size_t stride = ((size_t)n_slots * (hpp + 4) + 63) / 64 * 64;
    HIPCHECK(h, h->d_gnode.reserve(sizeof(float) * (stride * (size_t)B + 256)));
    P.gnode = h->d_gnode.as<float>();
    P.gnode_stride = (long long)stride;
  }
  if (gn8) {
    const size_t stride = (gnode_floats8(hpe, hpp, N) + 63) / 64 * 64;
    HIPCHECK(h, h->d_gnode.reserve(sizeof(float) * (stride * (size_t)B + 256)));  // (+ the staging copies' 1 KiB read granule)
    P.gnode = h->d_gnode.as<float>();
    P.gnode_stride = (long long)stride;
  }
  return GAUDI_OK;
}

// hpe / hpp: padded hidden sizes of the networks this call runs (0 = not used)
static int stage_graph(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, KParams& P, int hpe,
                       int hpp) {
  if (B <= 0 || N <= 0) return fail(h, GAUDI_E_INVALID, "B and N must be positive");
  // a node slot's row word holds molecule * N + node in 28 bits (sampler_kernel.h: row_of / comp_of)
  if ((int64_t)B * N >= (1 << 28)) return fail(h, GAUDI_E_CAPACITY, "B * N must stay below 2^28 per call: cut the request into several calls");
  h->run_variant = h->variant;
  h->run_split = 0;
  h->run_gn = false;
  h->run_two = false;
  h->run_gn8 = 0;
  h->run_pg = false;
  h->run_mr = false;
  h->run_groups = B;
  h->run_nslots = N;
  // a sin_embedding denoiser (24 edge features per first Linear, egnn_new.py:269-273) exists in the 4-wave family only
  const bool se = hpe && h->ecfg.sin_embedding;
  if (h->variant == 8 && !h->force_gn && !se) {
    const int rc8 = stage_graph8(h, B, N, node_mask, edge_mask, P, hpe, hpp);
    if (rc8 <= 0) return rc8;
    h->run_variant = 4;  // fall back to the 4-wave kernels for this call
    h->run_split = 0;
    P.pubx = P.pub_ch = 0;
  }
  h->run_variant = 4;
  // the 4-wave reverse pass parks [4 waves][N * 3] partial coordinate gradients in its 4 x 16 x (HP + 4) transposition scratch
  if (hpp && 4 * N * 3 > 4 * 16 * (hpp + 4))
    return fail(h, GAUDI_E_CAPACITY, "N too large for the 4-wave predictor kernels at this hidden size");
  Meta M;
  std::string err;
  int rc = build_meta(B, N, node_mask, edge_mask, M, err);
  if (rc) return fail(h, rc, err);
  HIPCHECK(h, h->d_mask.reserve(sizeof(float) * B * N));
  HIPCHECK(h, h->d_order.reserve(sizeof(int) * B));
  HIPCHECK(h, h->d_edges.reserve(sizeof(uint32_t) * M.edges.size()));
  HIPCHECK(h, h->d_emask.reserve(sizeof(float) * M.emask.size()));
  HIPCHECK(h, h->d_npairs.reserve(sizeof(int) * M.npairs.size()));
  HIPCHECK(h, h->d_seg.reserve(sizeof(uint32_t) * M.seg.size()));
  HIPCHECK(h, h->d_ncols.reserve(sizeof(int) * B));
  HIPCHECK(h, hipMemcpyAsync(h->d_ncols.p, M.ncols.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_mask.p, node_mask, sizeof(float) * B * N, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_order.p, M.order.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_edges.p, M.edges.data(), sizeof(uint32_t) * M.edges.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_emask.p, M.emask.data(), sizeof(float) * M.emask.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_npairs.p, M.npairs.data(), sizeof(int) * M.npairs.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_seg.p, M.seg.data(), sizeof(uint32_t) * M.seg.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));  // M goes out of scope
  P.B = B;
  P.N = N;
  P.NR = N;
  P.EW = M.EW;
  P.node_mask = h->d_mask.as<float>();
  P.order = h->d_order.as<int>();
  P.edges = h->d_edges.as<uint32_t>();
  P.emask = h->d_emask.as<float>();
  P.npairs = h->d_npairs.as<int>();
  P.seginfo = h->d_seg.as<uint32_t>();
  P.ncols = h->d_ncols.as<int>();
  // Molecules whose node buffers do not fit 160 KiB of LDS (beyond ~22 graph nodes at the default widths) run on the V4G
  // kernels: same code, node buffers in a per-workgroup global scratch (L2-resident).  The reference has no size cap
  // (sampling_edm.py:172-209); this one is N <= 255 (node indices are bytes in the edge words).
  // While the call is a shard of a larger logical batch (plan hint) or one of gaudi_sample's sub-batches, the choice is made
  // for the DENSE graph of N nodes, not for this batch's edges: the cuts of one logical batch must not land on different
  // kernels because one of them happens to be sparser.  A call that stands alone decides from its own edge lists (a sparse
  // hetero batch of 18-20 nodes fits the resident kernels where the dense graph would not).
  const int Dz = 3 + (hpe ? h->ecfg.in_node_nf : h->pcfg.in_nf);
  const bool part_of_batch = h->plan_min_slots || h->plan_force_waves || h->call_min_slots || h->call_force_waves || h->call_cut;
  const int ef = se ? 24 : 2;
  const bool have_g = se ? (pick_kernel_se(hpe, 0, true) && (!hpp || pick_kernel_g(0, hpp))) : have_kernels_g(hpe, hpp);
  if (h->force_gn || lds_bytes(hpe, hpp, N, Dz, part_of_batch ? std::max(M.EW, dense_ew4(N)) : M.EW, false, ef) > 160 * 1024) {
    if (have_g && lds_bytes(hpe, hpp, N, Dz, M.EW, true, ef) <= 160 * 1024) {
      h->run_gn = true;
      const size_t stride = (gnode_floats(hpe, hpp, N) + 63) / 64 * 64;
      HIPCHECK(h, h->d_gnode.reserve(sizeof(float) * stride * (size_t)B));
      P.gnode = h->d_gnode.as<float>();
      P.gnode_stride = (long long)stride;
    }
  }
  // Guided steps as two launches: the V4G kernels have no fused instantiation, and a sin_embedding denoiser has one at the tiny and the
  // default width pairs only -- every other width runs its denoiser-only kernel followed by the ordinary predictor-only kernel
  h->run_two = h->run_gn || (se && hpp && !pick_kernel_se(hpe, hpp, false));
  return GAUDI_OK;
}

static void fill_edm(gaudi_handle* h, KParams& P) {
  const gaudi_edm_config& c = h->ecfg;
  P.F = c.in_node_nf;
  P.T = c.diffusion_steps;
  P.edm.w = (h->variant == 8 && h->run_variant == 4) ? h->edm_w4.as<float>() : h->edm_w.as<float>();
  P.edm.w_bytes = (unsigned)h->edm_w_bytes;
  P.edm.F = c.in_node_nf;
  P.edm.L = c.n_layers;
  P.edm.S = c.inv_sublayers;
  P.edm.attention = c.attention;
  P.edm.use_tanh = c.tanh;
  P.edm.coords_range = c.coords_range;
  P.edm.norm_constant = c.norm_constant;
  // aggregation_method 'mean' (normalization_factor = 0 in the config): unsorted_segment_sum divides by the number of edges of
  // the dense list that share the row, masked ones included (egnn_new.py:416-420) = the call's padded node count
  P.edm.normf = c.normalization_factor > 0.f ? c.normalization_factor : (float)P.NR;
  P.edm.ktail = h->run_variant == 8 && has_ktail(c.hidden_nf, h->HPE);
  P.edm.ws = h->edm_ws.as<float>();
  P.edm.ws_bytes = (unsigned)h->edm_ws_bytes;
  P.edm.hinv = h->edm_hinv;
  P.coef = h->coef_d.as<float>();
  const float g0 = h->gamma[0];
  P.alpha0 = sqrtf(sigmoid_host(-g0));
  P.sigma0 = sqrtf(sigmoid_host(g0));
  P.sigma_x = expf(0.5f * g0);  // SNR(-0.5*gamma_0), en_diffusion.py:538
  P.nv0 = c.norm_values[0];
  P.nv1 = c.norm_values[1];
}

// -------------------------------------------------------------------------------------------------
extern "C" {

int gaudi_create(int device, gaudi_handle** out) {
  if (!out) return GAUDI_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return GAUDI_E_HIP;
  gaudi_handle* h = new gaudi_handle();
  h->device = device;
  if (const char* v = getenv("GAUDI_WAVES")) h->variant = atoi(v) == 4 ? 4 : 8;
  if (const char* v = getenv("GAUDI_EDGE_MATH")) h->split = std::string(v) != "fp32";
  if (const char* v = getenv("GAUDI_FORCE_GN")) h->force_gn = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_GN8")) h->gn8 = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_FORCE_GN8")) h->force_gn8 = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_FORCE_MR")) h->force_mr = atoi(v) != 0;  // diagnostic: one-round graphs on the MR kernels
  if (const char* v = getenv("GAUDI_PACK")) h->pack = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_GN8_PACK")) h->gn8_pack = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_GN8_PQ")) h->gn8_pq = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_WIDE_FULL")) h->wide_full = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_KEEP_H")) h->keep_h = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_FAMILY_SPLIT")) h->family_split = atoi(v) != 0;
  if (const char* v = getenv("GAUDI_PAIRS")) h->pairs = atoi(v);
  if (const char* v = getenv("GAUDI_PAIRS_CAP")) h->pairs_cap = atoi(v);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->num_cus = cus;
  }
  if (const char* v = getenv("GAUDI_PRED_ROUNDS")) h->pred_rounds = atoi(v) != 0;
  h->run_variant = h->variant;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return GAUDI_E_HIP;
  }
  *out = h;
  return GAUDI_OK;
}

void gaudi_destroy(gaudi_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->prof_log.reset(true);
  h->stab_log.reset(true);
  DevBuf* bufs[] = {&h->edm_w, &h->pred_w, &h->coef_d, &h->edm_w4, &h->pred_w4, &h->edm_ws, &h->pred_ws, &h->d_mask, &h->d_order, &h->d_edges, &h->d_emask, &h->d_npairs,
                    &h->d_seg, &h->d_zin, &h->d_zout, &h->d_t, &h->d_x, &h->d_h, &h->d_noise, &h->d_nan, &h->d_dpred,
                    &h->d_pred, &h->d_tw, &h->d_stash, &h->d_chain, &h->d_sx, &h->d_stype, &h->d_sn,
                    &h->d_sflags, &h->d_sdist, &h->d_sadj, &h->d_saux, &h->d_stab, &h->d_as, &h->d_ncols, &h->d_soff, &h->d_sidx,
                    &h->d_gnode, &h->d_rowmap, &h->d_compmol, &h->d_ncomp};
  for (DevBuf* b : bufs) b->release();
  h->p_pred.release();
  h->p_dpred.release();
  h->p_z.release();
  h->p_dz.release();
  h->d_dz.release();
  if (h->cb_event) (void)hipEventDestroy(h->cb_event);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* gaudi_last_error(const gaudi_handle* h) { return h ? h->err.c_str() : "null handle"; }
const char* gaudi_last_warning(const gaudi_handle* h) { return h ? h->warn.c_str() : ""; }
int gaudi_abi_version(void) { return GAUDI_ABI_VERSION; }
int gaudi_profile_clock(gaudi_handle* h, double* shader_mhz) {
  if (!h || !shader_mhz) return GAUDI_E_INVALID;
  *shader_mhz = 0.0;
  if (!h->d_clock.p) return GAUDI_OK;  // no profiled launch yet
  HIPCHECK(h, hipSetDevice(h->device));
  unsigned long long c[4] = {0, 0, 0, 0};
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  HIPCHECK(h, hipMemcpy(c, h->d_clock.p, sizeof(c), hipMemcpyDeviceToHost));
  if (c[3] > c[1] && c[2] > c[0]) *shader_mhz = 100.0 * (double)(c[2] - c[0]) / (double)(c[3] - c[1]);
  return GAUDI_OK;
}
int gaudi_last_keep_h(const gaudi_handle* h, int32_t* lds_floats) {
  if (!h || !lds_floats) return GAUDI_E_INVALID;
  *lds_floats = h->run_hk;
  return GAUDI_OK;
}
int gaudi_last_family_split(const gaudi_handle* h, int32_t* resident_molecules) {
  if (!h || !resident_molecules) return GAUDI_E_INVALID;
  *resident_molecules = h->last_split_resident;
  return GAUDI_OK;
}

int gaudi_load_edm(gaudi_handle* h, const gaudi_edm_config* cfg, int n, const char* const* names,
                   const float* const* tensors, const int64_t* numel) {
  if (!h || !cfg) return GAUDI_E_INVALID;
  HIPCHECK(h, hipSetDevice(h->device));
  const int H = cfg->hidden_nf, F = cfg->in_node_nf, F1 = F + 1, L = cfg->n_layers, S = cfg->inv_sublayers;
  if (H <= 0 || H > 256) return fail(h, GAUDI_E_INVALID, "hidden_nf must be in 1..256");
  if (F < 1 || F > 15) return fail(h, GAUDI_E_INVALID, "in_node_nf must be in 1..15");
  if (L < 1 || S < 1 || cfg->diffusion_steps < 1) return fail(h, GAUDI_E_INVALID, "bad n_layers/inv_sublayers/diffusion_steps");
  if (!(cfg->normalization_factor >= 0.f)) return fail(h, GAUDI_E_INVALID, "normalization_factor must be > 0 (or 0: 'mean' aggregation)");
  const int HP = round_hidden(H);
  if (!HP) return fail(h, GAUDI_E_INVALID, "no kernel instantiated for this hidden size");
  Tensors T;
  for (int i = 0; i < n; ++i) T.m[names[i]] = {tensors[i], numel[i]};
  const int EF = cfg->sin_embedding ? 24 : 2;  // edge_feat_nf (egnn_new.py:269-273): 2 x 12 sinusoids of (r, d0), or the two scalars
  EdmLayout lay{HP, F1, L, S, EF};
  const std::string p = "dynamics.egnn.";
  const int PK = HP * HP;
  // tile layout: lane-linear for the 8-wave kernels, row-major for the 4-wave ones (kept as the fallback of the 8-wave
  // variant for graphs that do not fit it)
  // the exponent of the fp16-pair node images: every node-GEMM matrix of the network (the A | B column blocks of W1, Wn1, Wn2)
  float hscale = 0.f;
  {
    NodeScale ns;
    for (int l = 0; l < L; ++l) {
      for (int s = 0; s <= S; ++s) {
        const std::string q = p + "e_block_" + std::to_string(l) + (s < S ? ".gcl_" + std::to_string(s) + "." : ".gcl_equiv.");
        ns.see(T.peek(q + (s < S ? "edge_mlp.0.weight" : "coord_mlp.0.weight"), (int64_t)H * (2 * H + EF)), H, 2 * H + EF, 0, 2 * H);
        ns.see(T.peek(q + (s < S ? "edge_mlp.2.weight" : "coord_mlp.2.weight"), (int64_t)H * H), H, H, 0, H, true);
        if (s < S) {
          ns.see(T.peek(q + "node_mlp.0.weight", (int64_t)H * 2 * H), H, 2 * H, 0, 2 * H);
          ns.see(T.peek(q + "node_mlp.2.weight", (int64_t)H * H), H, H, 0, H);
        }
      }
    }
    hscale = ns.scale();
    if (!(hscale > 0.f) && h->variant == 8 && h->split && EF == 2)
      h->warn = std::string("EDM weights: the fp16-pair images cannot carry this weight set (") + ns.why() +
                "): its calls run the fp32-instruction kernels (about 0.55 x the speed, same results at fp32 accuracy)";
  }
  auto pack = [&](bool lane_linear, std::vector<float>& w, std::vector<float>* ws) {
  w.assign((size_t)lay.total(), 0.f);
  if (ws) ws->assign(2 * (size_t)lay.total(), 0.f);
  // only the matrices of the edge-level GEMMs carry the K tail: pe packs those, pn the node-level ones
  PackMode pe;
  pe.lane_linear = lane_linear;
  pe.ktail = lane_linear && has_ktail(H, HP);
  pe.wbase = w.data();
  pe.ws = ws;
  pe.hscale = ws ? hscale : 0.f;
  pe.hktail = has_ktail(H, HP);
  const PackMode pn = pe.with_ktail(false);
  {
    const float* ew = T.get(p + "embedding.weight", (int64_t)H * F1);
    const float* eb = T.get(p + "embedding.bias", H);
    const float* ow = T.get(p + "embedding_out.weight", (int64_t)F1 * H);
    const float* ob = T.get(p + "embedding_out.bias", F1);
    if (ew && eb && ow && ob) {
      for (int f = 0; f < H; ++f)
        for (int k = 0; k < F1; ++k) w[lay.emb_w() + f * F1 + k] = ew[f * F1 + k];
      pack_vec(&w[lay.emb_b()], eb, H);
      for (int o = 0; o < F1; ++o)
        for (int k = 0; k < H; ++k) w[lay.out_w() + o * HP + k] = ow[o * H + k];
      pack_vec(&w[lay.out_b()], ob, F1);
    }
  }
  const int ld1 = 2 * H + EF;
  for (int l = 0; l < L; ++l) {
    for (int s = 0; s < S; ++s) {
      const std::string q = p + "e_block_" + std::to_string(l) + ".gcl_" + std::to_string(s) + ".";
      float* G = &w[lay.gcl(l, s)];
      float* V = G + 6 * PK;
      const float* W1 = T.get(q + "edge_mlp.0.weight", (int64_t)H * ld1);
      const float* b1 = T.get(q + "edge_mlp.0.bias", H);
      const float* W2 = T.get(q + "edge_mlp.2.weight", (int64_t)H * H);
      const float* b2 = T.get(q + "edge_mlp.2.bias", H);
      const float* Wn1 = T.get(q + "node_mlp.0.weight", (int64_t)H * 2 * H);
      const float* bn1 = T.get(q + "node_mlp.0.bias", H);
      const float* Wn2 = T.get(q + "node_mlp.2.weight", (int64_t)H * H);
      const float* bn2 = T.get(q + "node_mlp.2.bias", H);
      const float *wa = nullptr, *ba = nullptr;
      if (cfg->attention) {
        wa = T.get(q + "att_mlp.0.weight", H);
        ba = T.get(q + "att_mlp.0.bias", 1);
      }
      if (!(W1 && b1 && W2 && b2 && Wn1 && bn1 && Wn2 && bn2) || (cfg->attention && !(wa && ba))) continue;
      pack_node_matrix(pn, G, W1, H, ld1, 0, HP);
      pack_node_matrix(pn, G + PK, W1, H, ld1, H, HP);
      pack_edge_matrix(pe, G + 2 * PK, W2, H, H, 0, HP);
      pack_node_matrix(pn, G + 3 * PK, Wn1, H, 2 * H, 0, HP);
      pack_node_matrix(pn, G + 4 * PK, Wn1, H, 2 * H, H, HP);
      pack_node_matrix(pn, G + 5 * PK, Wn2, H, H, 0, HP);
      for (int k = 0; k < EF; ++k) pack_col(V + k * HP, W1, H, ld1, 2 * H + k);  // c_r, c_d (or the 24 sinusoid columns)
      pack_vec(V + EF * HP, b1, H);
      pack_vec(V + (EF + 1) * HP, b2, H);
      if (wa) pack_vec(V + (EF + 2) * HP, wa, H);
      pack_vec(V + (EF + 3) * HP, bn1, H);
      pack_vec(V + (EF + 4) * HP, bn2, H);
      if (ba) V[(EF + 5) * HP] = ba[0];
      V[(EF + 5) * HP + 1] = col_absmax(W1, H, ld1, 2 * H);      // max |c_r|, max |c_d|: the split edge GEMMs' column scales are
      V[(EF + 5) * HP + 2] = col_absmax(W1, H, ld1, 2 * H + 1);  // bounded with them (w8_split.h; EF = 2 only)
    }
    const std::string q = p + "e_block_" + std::to_string(l) + ".gcl_equiv.";
    float* E = &w[lay.equ(l)];
    float* V = E + 3 * PK;
    const float* W1 = T.get(q + "coord_mlp.0.weight", (int64_t)H * ld1);
    const float* b1 = T.get(q + "coord_mlp.0.bias", H);
    const float* W2 = T.get(q + "coord_mlp.2.weight", (int64_t)H * H);
    const float* b2 = T.get(q + "coord_mlp.2.bias", H);
    const float* w3 = T.get(q + "coord_mlp.4.weight", H);
    if (!(W1 && b1 && W2 && b2 && w3)) continue;
    pack_node_matrix(pn, E, W1, H, ld1, 0, HP);
    pack_node_matrix(pn, E + PK, W1, H, ld1, H, HP);
    pack_edge_matrix(pe, E + 2 * PK, W2, H, H, 0, HP);
    for (int k = 0; k < EF; ++k) pack_col(V + k * HP, W1, H, ld1, 2 * H + k);
    pack_vec(V + EF * HP, b1, H);
    pack_vec(V + (EF + 1) * HP, b2, H);
    pack_vec(V + (EF + 2) * HP, w3, H);
    V[(EF + 3) * HP] = col_absmax(W1, H, ld1, 2 * H);
    V[(EF + 3) * HP + 1] = col_absmax(W1, H, ld1, 2 * H + 1);
  }
  };
  std::vector<float> w, ws;
  // a sin_embedding denoiser only ever runs on the 4-wave kernels (stage_graph): no fp16-pair images, and the row-major pack twice
  const bool want_split = h->variant == 8 && h->split && EF == 2;
  pack(h->variant == 8 && EF == 2, w, want_split ? &ws : nullptr);
  if (!T.missing.empty()) return fail(h, GAUDI_E_MISSING, "EDM checkpoint tensor missing or mis-shaped: " + T.missing);
  HIPCHECK(h, h->edm_w.reserve(sizeof(float) * w.size()));
  HIPCHECK(h, hipMemcpy(h->edm_w.p, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
  h->edm_w_bytes = sizeof(float) * w.size();
  h->edm_ws_bytes = 0;
  if (want_split) {
    HIPCHECK(h, h->edm_ws.reserve(sizeof(float) * ws.size()));
    HIPCHECK(h, hipMemcpy(h->edm_ws.p, ws.data(), sizeof(float) * ws.size(), hipMemcpyHostToDevice));
    h->edm_ws_bytes = sizeof(float) * ws.size();
  }
  h->edm_hinv = want_split && hscale > 0.f ? 1.0f / hscale : 0.f;
  if (h->variant == 8) {
    pack(false, w, nullptr);
    HIPCHECK(h, h->edm_w4.reserve(sizeof(float) * w.size()));
    HIPCHECK(h, hipMemcpy(h->edm_w4.p, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
  }
  h->gamma = make_gamma(cfg->diffusion_steps, cfg->noise_power, cfg->noise_precision);
  make_coef(h->gamma, cfg->diffusion_steps, h->coef);
  HIPCHECK(h, h->coef_d.reserve(sizeof(float) * h->coef.size()));
  HIPCHECK(h, hipMemcpy(h->coef_d.p, h->coef.data(), sizeof(float) * h->coef.size(), hipMemcpyHostToDevice));
  h->ecfg = *cfg;
  h->HPE = HP;
  h->has_edm = true;
  return GAUDI_OK;
}

int gaudi_get_gamma(gaudi_handle* h, float* gamma_out) {
  if (!h || !gamma_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  std::memcpy(gamma_out, h->gamma.data(), sizeof(float) * h->gamma.size());
  return GAUDI_OK;
}

int gaudi_get_step_coefficients(gaudi_handle* h, float* coef_out) {
  if (!h || !coef_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  std::memcpy(coef_out, h->coef.data(), sizeof(float) * h->coef.size());
  return GAUDI_OK;
}

int gaudi_phi(gaudi_handle* h, int B, int N, const float* z, const float* t, const float* node_mask,
              const float* edge_mask, float* eps_out) {
  if (!h || !z || !t || !node_mask || !edge_mask || !eps_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  HIPCHECK(h, hipSetDevice(h->device));
  KParams P{};
  int rc = stage_graph(h, B, N, node_mask, edge_mask, P, h->HPE, 0);
  if (rc) return rc;
  fill_edm(h, P);
  const size_t zb = sizeof(float) * B * N * (3 + P.F);
  HIPCHECK(h, h->d_zin.reserve(zb));
  HIPCHECK(h, h->d_zout.reserve(zb));
  HIPCHECK(h, h->d_t.reserve(sizeof(float) * B));
  HIPCHECK(h, hipMemcpyAsync(h->d_zin.p, z, zb, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(h, hipMemcpyAsync(h->d_t.p, t, sizeof(float) * B, hipMemcpyHostToDevice, h->stream));
  P.mode = MODE_PHI;
  P.z_in = h->d_zin.as<float>();
  P.z_out = h->d_zout.as<float>();
  P.t_in = h->d_t.as<float>();
  rc = launch(h, P, h->HPE, 0, 0);
  if (rc) return rc;
  HIPCHECK(h, hipMemcpyAsync(eps_out, h->d_zout.p, zb, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return GAUDI_OK;
}

static int fill_pred(gaudi_handle* h, KParams& P, const float* target_w, int B, int N);  // pred_host.inc

// diagnostics replacing assert_correctly_masked / assert_mean_zero_with_mask (utils.py:52-65) and the
// CoG re-projection of en_diffusion.py:1000-1006 (batch-wide condition -> host side)
static void finish_sample(int B, int N, const float* node_mask, float* x_out, int nanc, gaudi_diag* diag) {
  float leak = 0.f, cog = 0.f, big = 0.f;
  for (int b = 0; b < B; ++b) {
    float s[3] = {0, 0, 0};
    for (int n = 0; n < N; ++n)
      for (int d = 0; d < 3; ++d) {
        const float v = x_out[((size_t)b * N + n) * 3 + d];
        s[d] += v;
        big = std::max(big, std::fabs(v));
        leak = std::max(leak, std::fabs(v * (1.f - node_mask[b * N + n])));
      }
    for (int d = 0; d < 3; ++d) cog = std::max(cog, std::fabs(s[d]));
  }
  int reproj = 0;
  if (cog > 5e-2f) {
    reproj = 1;
    for (int b = 0; b < B; ++b) {
      float cnt = 0.f;
      for (int n = 0; n < N; ++n) cnt += node_mask[b * N + n];
      cnt = std::max(cnt, 1.f);
      for (int d = 0; d < 3; ++d) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += x_out[((size_t)b * N + n) * 3 + d];
        const float mean = s / cnt;
        for (int n = 0; n < N; ++n) x_out[((size_t)b * N + n) * 3 + d] -= mean * node_mask[b * N + n];
      }
    }
  }
  if (diag) {
    diag->max_masked_leak = leak;
    diag->max_cog_abs = cog;
    diag->max_cog_rel = cog / (big + 1e-10f);
    diag->nan_count = nanc;
    diag->reprojected = reproj;
  }
}

// shared by gaudi_step / gaudi_decode / gaudi_sample
static int run_chain(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, const float* z_in,
                     bool do_init, int s_hi, int s_lo, bool do_decode, const float* noise, int draw_base, int n_draws,
                     uint64_t seed, int64_t sample_offset, float std0, const float* target_w, float scale,
                     float* z_out, float* x_out, float* onehot_out, int* nan_count, float* chain_out = nullptr,
                     int keep_frames = 0) {
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  if (target_w && !h->has_pred) return fail(h, GAUDI_E_STATE, "guided sampling needs predictor weights");
  HIPCHECK(h, hipSetDevice(h->device));
  KParams P{};
  h->pack_now = chain_out == nullptr;  // sampling calls may pack small molecules into one workgroup (stage_graph8)
  int rc = stage_graph(h, B, N, node_mask, edge_mask, P, h->HPE, target_w ? h->HPP : 0);
  h->pack_now = false;
  if (rc) return rc;
  fill_edm(h, P);
  const int D = 3 + P.F, T = P.T;
  if (s_hi >= T || s_lo < 0) return fail(h, GAUDI_E_INVALID, "step index out of range");
  const size_t zb = sizeof(float) * B * N * D;
  HIPCHECK(h, h->d_zin.reserve(zb));
  HIPCHECK(h, h->d_zout.reserve(zb));
  HIPCHECK(h, h->d_x.reserve(sizeof(float) * B * N * 3));
  HIPCHECK(h, h->d_h.reserve(sizeof(float) * B * N * P.F));
  HIPCHECK(h, h->d_nan.reserve(sizeof(int)));
  HIPCHECK(h, hipMemsetAsync(h->d_nan.p, 0, sizeof(int), h->stream));
  if (P.rowmap != nullptr) {  // packed: masked nodes have no slot in any workgroup -- their rows stay zero
    HIPCHECK(h, hipMemsetAsync(h->d_zin.p, 0, zb, h->stream));
    HIPCHECK(h, hipMemsetAsync(h->d_zout.p, 0, zb, h->stream));
    HIPCHECK(h, hipMemsetAsync(h->d_x.p, 0, sizeof(float) * B * N * 3, h->stream));
    HIPCHECK(h, hipMemsetAsync(h->d_h.p, 0, sizeof(float) * B * N * P.F, h->stream));
  }
  if (z_in) HIPCHECK(h, hipMemcpyAsync(h->d_zin.p, z_in, zb, hipMemcpyHostToDevice, h->stream));
  const bool fixn = h->fix_noise && do_init;  // whole-chain calls only (gaudi_step / gaudi_decode inject per-molecule draws)
  const size_t nzb = fixn ? sizeof(float) * N * D : zb;  // bytes of one raw draw
  if (noise) {
    HIPCHECK(h, h->d_noise.reserve(nzb * (size_t)n_draws));
    HIPCHECK(h, hipMemcpyAsync(h->d_noise.p, noise, nzb * (size_t)n_draws, hipMemcpyHostToDevice, h->stream));
    P.noise = h->d_noise.as<float>();
  }
  P.draw_base = draw_base;
  P.draw_stride = fixn ? (long long)N * D : (long long)B * N * D;
  P.fix_noise = fixn ? 1 : 0;
  P.fix_key = h->fix_key;
  P.seed = seed;
  P.sample_offset = sample_offset;
  P.std0 = std0;
  P.mode = MODE_SAMPLE;
  P.x_out = h->d_x.as<float>();
  P.h_out = h->d_h.as<float>();
  P.nan_count = h->d_nan.as<int>();
  P.guided = target_w != nullptr;
  P.scale = scale;
  if (chain_out) {
    HIPCHECK(h, h->d_chain.reserve(zb * (size_t)keep_frames));
    HIPCHECK(h, hipMemsetAsync(h->d_chain.p, 0, zb * (size_t)keep_frames, h->stream));
    P.chain_out = h->d_chain.as<float>();
    P.keep_frames = keep_frames;
  }
  int hpp = 0;
  if (target_w) {
    rc = fill_pred(h, P, target_w, B, N);
    if (rc) return rc;
    hpp = h->HPP;
  }
  float* zin = h->d_zin.as<float>();
  float* zout = h->d_zout.as<float>();
  if (h->run_two && target_w) {
    // Large molecules (V4G kernels) and sin_embedding denoisers without a fused kernel, guided: every reverse step is two launches -- the EDM-only kernel runs the step up to
    // z_s before guidance (split = 1: denoise, update with noise), the predictor-only kernel the guidance update, the
    // projection and the NaN scrub (MODE_GUIDE) -- then one decode pass.
    for (int s = s_hi; s >= s_lo; --s) {
      P.mode = MODE_SAMPLE;
      P.s_hi = P.s_lo = s;
      P.do_init = (s == s_hi) && do_init;
      P.do_decode = 0;
      P.split = 1;
      P.z_in = zin;
      P.z_out = zout;
      rc = launch(h, P, h->HPE, 0, 1);
      if (rc) return rc;
      P.mode = MODE_GUIDE;
      P.do_init = 0;
      P.split = 0;
      P.z_in = zout;
      P.z_out = zin;
      rc = launch(h, P, 0, hpp, 0);
      if (rc) return rc;
    }
    if (do_decode) {
      P.mode = MODE_SAMPLE;
      P.split = 0;
      P.s_hi = -1;
      P.s_lo = 0;
      P.do_init = (s_hi < s_lo) && do_init;
      P.do_decode = 1;
      P.z_in = zin;
      P.z_out = zout;
      rc = launch(h, P, h->HPE, 0, 0);
      if (rc) return rc;
      std::swap(zin, zout);
    }
  } else {
  // chunk the chain into launches of steps_per_launch steps; z ping-pongs zout -> zin
  bool first = true;
  int s = s_hi;
  const bool any_steps = s_hi >= s_lo;
  do {
    const int lo = any_steps ? std::max(s_lo, s - h->steps_per_launch + 1) : s + 1;
    P.s_hi = s;
    P.s_lo = lo;
    P.do_init = first && do_init;
    P.do_decode = do_decode && lo <= s_lo;
    P.z_in = zin;
    P.z_out = zout;
    rc = launch(h, P, h->HPE, hpp, any_steps ? (s - lo + 1) : 0);
    if (rc) return rc;
    std::swap(zin, zout);
    first = false;
    s = lo - 1;
  } while (any_steps && s >= s_lo);
  }
  // after the swap, `zin` holds the latest z
  if (z_out) HIPCHECK(h, hipMemcpyAsync(z_out, zin, zb, hipMemcpyDeviceToHost, h->stream));
  if (do_decode) {
    HIPCHECK(h, hipMemcpyAsync(x_out, h->d_x.p, sizeof(float) * B * N * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipMemcpyAsync(onehot_out, h->d_h.p, sizeof(float) * B * N * P.F, hipMemcpyDeviceToHost, h->stream));
  }
  if (chain_out)
    HIPCHECK(h, hipMemcpyAsync(chain_out, h->d_chain.p, zb * (size_t)keep_frames, hipMemcpyDeviceToHost, h->stream));
  int nanc = 0;
  HIPCHECK(h, hipMemcpyAsync(&nanc, h->d_nan.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  if (nan_count) *nan_count = nanc;
  return GAUDI_OK;
}

int gaudi_step(gaudi_handle* h, int B, int N, int s_idx, const float* z_t, const float* node_mask,
               const float* edge_mask, const float* eps_raw, const float* target_w, float scale, float* zs_out) {
  if (!h || !z_t || !node_mask || !edge_mask || !eps_raw || !zs_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  const int T = h->ecfg.diffusion_steps;
  return run_chain(h, B, N, node_mask, edge_mask, z_t, false, s_idx, s_idx, false, eps_raw, T - s_idx, 1, 0, 0, 1.0f,
                   target_w, scale, zs_out, nullptr, nullptr, nullptr);
}

int gaudi_decode(gaudi_handle* h, int B, int N, const float* z0, const float* node_mask, const float* edge_mask,
                 const float* eps_raw, float* x_out, float* onehot_out) {
  if (!h || !z0 || !node_mask || !edge_mask || !eps_raw || !x_out || !onehot_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  const int T = h->ecfg.diffusion_steps;
  return run_chain(h, B, N, node_mask, edge_mask, z0, false, -1, 0, true, eps_raw, T + 1, 1, 0, 0, 1.0f, nullptr, 0.f,
                   nullptr, x_out, onehot_out, nullptr);
}

// Largest sub-batch one chain may run at once.  The guided path keeps an activation stash of 2.9 MB per molecule (default
// sizes) for the whole call; very large requests are cut into sub-batches of whole multiples of 256 molecules (one per
// CU) that fit `budget` bytes.  Noise is keyed by the global sample index, so the result does not depend on the cut.
static int max_sub_batch(gaudi_handle* h, int B, int N, bool guided) {
  if (!guided) return B;
  const long long per_mol = 4LL * pred_stash_floats(N, h->HPP, h->pcfg.n_layers, dense_ew4(N));
  long long budget = 0;
  if (const char* e = getenv("GAUDI_MAX_WORKSPACE_MB")) budget = atoll(e) * (1LL << 20);
  const bool forced = budget > 0;
  if (!forced) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return B;
    budget = (long long)((free_b + h->d_stash.cap) * 0.8);
  }
  long long bmax = std::max(1LL, budget / std::max(1LL, per_mol));
  if (bmax >= 256) bmax = bmax / 256 * 256;
  else if (!forced) bmax = std::min<long long>(B, 256);  // let the allocation itself report a too-small device
  return (int)std::min<long long>(B, bmax);
}

// Node slots of the widest packed group the RESIDENT split kernels take at these widths (one round of eight edge tiles), below N;
// 0: none (no split images, no such kernel).  A function of the widths only.
static int resident_node_limit(gaudi_handle* h, int N, bool guided) {
  const int hpe = h->HPE, hpp = guided ? h->HPP : 0;
  if (h->variant != 8 || !h->split || !h->edm_ws_bytes || (hpp && !h->pred_ws_bytes)) return 0;
  if (GAUDI_NODE_F16 && (!(h->edm_hinv > 0.f) || (hpp && !(h->pred_hinv > 0.f)))) return 0;
  const int Dz = 3 + h->ecfg.in_node_nf;
  for (int ng = std::min(N - 1, 32); ng >= 8; --ng)
    for (int mode = 1; mode <= 2; ++mode) {
      int pubx = 0, pub_ch = 0;
      if (pick_kernel8_mode(hpe, hpp, mode, false) && plan_pub8(hpe, hpp, ng, Dz, 16 * w8::kWaves, mode, pubx, pub_ch)) return ng;
    }
  return 0;
}

int gaudi_sample(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                 int64_t sample_offset, const float* noise, float std, const float* target_w, float scale,
                 float* x_out, float* onehot_out, float* z0_out, gaudi_diag* diag) {
  if (!h || !node_mask || !edge_mask || !x_out || !onehot_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  if (target_w && !h->has_pred) return fail(h, GAUDI_E_STATE, "guided sampling needs predictor weights");
  if (B <= 0 || N <= 0) return fail(h, GAUDI_E_INVALID, "B and N must be positive");
  HIPCHECK(h, hipSetDevice(h->device));
  const int T = h->ecfg.diffusion_steps, F = h->ecfg.in_node_nf, D = 3 + F;
  struct CallHint {
    gaudi_handle* h;
    ~CallHint() {
      h->call_min_slots = h->call_force_waves = h->call_narrow = 0;
      h->call_cut = false;
      h->call_molmap = nullptr;
    }
  } call_hint{h};
  h->last_split_resident = 0;
  // ---- per-molecule kernel family (round 6).  A request whose padded N is beyond the resident kernels' LDS limit used to run
  // EVERY molecule on the V8G kernels (node buffers in a global scratch: 20 % slower on a molecule that would fit).
  // Now the molecules that fit the resident kernels on their own -- at most `lim` live nodes and one round of eight edge tiles: a
  // function of the molecule's own graph and the widths, so a molecule's kernel (and its rounding) does not depend on which
  // other molecules share the call or the shard -- form a first bucket that runs packed on the resident kernels; the rest
  // run on V8G as before.  Noise is keyed by the molecule's index in the request either way.
  std::vector<int32_t> small, large;
  int lim = 0;
  if (h->variant == 8 && h->family_split && h->pack && h->gn8 && h->gn8_pack && !h->force_gn && !h->force_gn8 && !h->fix_noise &&
      !h->plan_force_waves && (int64_t)B * N < (1 << 28)) {
    Meta8 M;
    std::string err;
    const int hpp = target_w ? h->HPP : 0;
    if (build_meta8(B, N, node_mask, edge_mask, M, err, h->plan_min_slots) == GAUDI_OK) {
      int pubx = 0, pub_ch = 0;
      bool fits = false;  // the unpacked resident plan of the whole call, any split mode
      for (int mode = 1; mode <= 2 && !fits; ++mode)
        fits = pick_kernel8_mode(h->HPE, hpp, mode, hpp && M.S > 16 * w8::kWaves) && plan_pub8(h->HPE, hpp, N, D, M.S, mode, pubx, pub_ch);
      if (!fits && (lim = resident_node_limit(h, N, target_w != nullptr)) > 0) {
        const std::vector<std::vector<int>> used = used_nodes(B, N, node_mask, edge_mask);
        for (int b = 0; b < B; ++b) ((int)used[b].size() <= lim && M.ntiles[b] <= w8::kWaves ? small : large).push_back(b);
      }
    }
  }
  struct Bucket {
    const std::vector<int32_t>* idx;  // nullptr: the whole request in place
    int narrow;
  };
  std::vector<Bucket> buckets;
  if (small.empty()) buckets.push_back({nullptr, 0});
  else {
    buckets.push_back({&small, lim});
    if (!large.empty()) buckets.push_back({&large, 0});
    h->last_split_resident = (int)small.size();
  }
  int nanc = 0;
  std::vector<float> nz, gm, ge, gx, gh, gz;
  for (const Bucket& bk : buckets) {
    const int Bb = bk.idx ? (int)bk.idx->size() : B;
    const float *nmb = node_mask, *emb = edge_mask, *nsb = noise;
    float *xo = x_out, *ho = onehot_out, *zo = z0_out;
    if (bk.idx) {  // gather the bucket's molecules
      gm.resize((size_t)Bb * N);
      ge.resize((size_t)Bb * N * N);
      gx.assign((size_t)Bb * N * 3, 0.f);
      gh.assign((size_t)Bb * N * F, 0.f);
      for (int k = 0; k < Bb; ++k) {
        const int b = (*bk.idx)[k];
        std::memcpy(&gm[(size_t)k * N], node_mask + (size_t)b * N, sizeof(float) * N);
        std::memcpy(&ge[(size_t)k * N * N], edge_mask + (size_t)b * N * N, sizeof(float) * N * N);
      }
      nmb = gm.data();
      emb = ge.data();
      xo = gx.data();
      ho = gh.data();
      if (z0_out) {
        gz.assign((size_t)Bb * N * D, 0.f);
        zo = gz.data();
      }
    }
    h->call_narrow = bk.narrow;
    h->call_min_slots = h->call_force_waves = 0;
    const int bmax = max_sub_batch(h, Bb, N, target_w != nullptr);
    // sub-batches plan with the whole batch's graph figures (same kernel family and edge-GEMM arithmetic for every cut)
    h->call_cut = bmax < Bb;
    if (bmax < Bb && h->variant == 8 && !bk.narrow) {
      Meta8 M;
      std::string err;
      const int rc = build_meta8(Bb, N, nmb, emb, M, err);
      if (rc == GAUDI_E_CAPACITY) h->call_force_waves = 4;
      else if (rc) return fail(h, rc, err);
      else h->call_min_slots = M.S;
    }
    for (int b0 = 0; b0 < Bb; b0 += bmax) {
      const int nb = std::min(bmax, Bb - b0);
      const float* nzp = nsb;
      if (noise && !h->fix_noise && (nb != B || bk.idx)) {  // gather this sub-batch's draws out of [T+2][B][N][D]
        nz.resize((size_t)(T + 2) * nb * N * D);
        for (int d = 0; d < T + 2; ++d)
          for (int k = 0; k < nb; ++k) {
            const int b = bk.idx ? (*bk.idx)[b0 + k] : b0 + k;
            std::memcpy(&nz[((size_t)d * nb + k) * N * D], noise + ((size_t)d * B + b) * N * D, sizeof(float) * (size_t)N * D);
          }
        nzp = nz.data();
      }
      h->call_molmap = bk.idx ? bk.idx->data() + b0 : nullptr;
      int nan_sub = 0;
      int rc = run_chain(h, nb, N, nmb + (size_t)b0 * N, emb + (size_t)b0 * N * N, nullptr, true, T - 1, 0, true, nzp, 0, T + 2, seed,
                         bk.idx ? sample_offset : sample_offset + b0, std, target_w, scale, zo ? zo + (size_t)b0 * N * D : nullptr,
                         xo + (size_t)b0 * N * 3, ho + (size_t)b0 * N * F, &nan_sub);
      h->call_molmap = nullptr;
      if (rc) return rc;
      nanc += nan_sub;
    }
    if (bk.idx)  // scatter the bucket's results
      for (int k = 0; k < Bb; ++k) {
        const int b = (*bk.idx)[k];
        std::memcpy(x_out + (size_t)b * N * 3, &gx[(size_t)k * N * 3], sizeof(float) * N * 3);
        std::memcpy(onehot_out + (size_t)b * N * F, &gh[(size_t)k * N * F], sizeof(float) * N * F);
        if (z0_out) std::memcpy(z0_out + (size_t)b * N * D, &gz[(size_t)k * N * D], sizeof(float) * N * D);
      }
  }
  finish_sample(B, N, node_mask, x_out, nanc, diag);
  return GAUDI_OK;
}

}  // extern "C"

// gaudi_sample_cb / gaudi_sample_cbz: exactly one of the two callbacks is set
static int sample_cb_impl(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                          int64_t sample_offset, const float* noise, float std, gaudi_target_cb target_grad,
                          gaudi_target_cbz target_grad_z, void* user, float scale, float* x_out, float* onehot_out, float* z0_out,
                          gaudi_diag* diag) {
  if (!h || !node_mask || !edge_mask || !x_out || !onehot_out || (!target_grad && !target_grad_z)) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  if (!h->has_pred) return fail(h, GAUDI_E_STATE, "guided sampling needs predictor weights");
#ifdef GAUDI_STAMPS
  // the stamped diagnostic build times the fused step only: its 8-wave guide phase drops the direct dT/dz term (sampler_kernel.h)
  if (target_grad_z) return fail(h, GAUDI_E_INVALID, "gaudi_sample_cbz is not available in the GAUDI_STAMPS diagnostic build");
#endif
  HIPCHECK(h, hipSetDevice(h->device));
  KParams P{};
  int rc = stage_graph(h, B, N, node_mask, edge_mask, P, h->HPE, h->HPP);
  if (rc) return rc;
  fill_edm(h, P);
  const int D = 3 + P.F, T = P.T, K = h->pcfg.out_nf;
  const size_t zb = sizeof(float) * B * N * D, pb = sizeof(float) * B * K;
  HIPCHECK(h, h->d_zin.reserve(zb));
  HIPCHECK(h, h->d_zout.reserve(zb));
  HIPCHECK(h, h->d_x.reserve(sizeof(float) * B * N * 3));
  HIPCHECK(h, h->d_h.reserve(sizeof(float) * B * N * P.F));
  HIPCHECK(h, h->d_nan.reserve(sizeof(int)));
  HIPCHECK(h, h->d_pred.reserve(pb));
  HIPCHECK(h, h->d_dpred.reserve(pb));
  HIPCHECK(h, h->p_pred.reserve(pb));
  HIPCHECK(h, h->p_dpred.reserve(pb));
  if (target_grad_z) {  // the target also depends on z directly: z_s goes to the host, scale * mask * dT/dz comes back
    HIPCHECK(h, h->d_dz.reserve(zb));
    HIPCHECK(h, h->p_z.reserve(zb));
    HIPCHECK(h, h->p_dz.reserve(zb));
  }
  if (!h->cb_event) HIPCHECK(h, hipEventCreateWithFlags(&h->cb_event, hipEventDisableTiming));
  HIPCHECK(h, hipMemsetAsync(h->d_nan.p, 0, sizeof(int), h->stream));
  const size_t nzb = h->fix_noise ? sizeof(float) * N * D : zb;
  if (noise) {
    HIPCHECK(h, h->d_noise.reserve(nzb * (size_t)(T + 2)));
    HIPCHECK(h, hipMemcpyAsync(h->d_noise.p, noise, nzb * (size_t)(T + 2), hipMemcpyHostToDevice, h->stream));
    P.noise = h->d_noise.as<float>();
  }
  P.draw_base = 0;
  P.draw_stride = h->fix_noise ? (long long)N * D : (long long)B * N * D;
  P.fix_noise = h->fix_noise ? 1 : 0;
  P.fix_key = h->fix_key;
  P.seed = seed;
  P.sample_offset = sample_offset;
  P.std0 = std;
  P.mode = MODE_SAMPLE;
  P.x_out = h->d_x.as<float>();
  P.h_out = h->d_h.as<float>();
  P.nan_count = h->d_nan.as<int>();
  P.guided = 1;
  P.scale = scale;
  std::vector<float> zero_w(16, 0.f);
  rc = fill_pred(h, P, zero_w.data(), B, N);
  if (rc) return rc;
  P.pred_out = h->d_pred.as<float>();
  P.dpred_in = h->d_dpred.as<float>();
  float* pred = h->p_pred.as<float>();
  float* dT = h->p_dpred.as<float>();
  float* zin = h->d_zin.as<float>();
  float* zout = h->d_zout.as<float>();
  // Large molecules (V4G kernels: node buffers in global memory) have no fused EDM + predictor instantiation (DESIGN.md
  // 7.12): phase A is the EDM-only kernel (split = 1: z_t -> z_s before guidance) followed by the predictor-only kernel's
  // forward half (MODE_GUIDE, split = 1), phase B the predictor-only kernel's second half (MODE_GUIDE, split = 2).
  const bool gn = h->run_two;
  // GAUDI_DEBUG_CB: where a callback step's host time goes (enqueue / wait for pred / the caller's function)
  const bool dbg_cb = getenv("GAUDI_DEBUG_CB") != nullptr;
  double t_enq = 0, t_wait = 0, t_user = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int s = T - 1; s >= 0; --s) {
    const double t0 = dbg_cb ? now() : 0;
    // phase A: z_t -> z_s (before guidance) and pred = predictor(z_s, t); the activation stash stays on the device
    P.mode = MODE_SAMPLE;
    P.s_hi = P.s_lo = s;
    P.do_init = s == T - 1;
    P.do_decode = 0;
    P.split = 1;
    P.z_in = zin;
    P.z_out = zout;
    rc = launch(h, P, h->HPE, gn ? 0 : h->HPP, 1);
    if (rc) return rc;
    if (gn) {
      P.mode = MODE_GUIDE;
      P.do_init = 0;
      P.z_in = zout;
      P.z_out = zin;
      rc = launch(h, P, 0, h->HPP, 0);
      if (rc) return rc;
    }
    HIPCHECK(h, hipMemcpyAsync(pred, h->d_pred.p, pb, hipMemcpyDeviceToHost, h->stream));
    if (target_grad_z) HIPCHECK(h, hipMemcpyAsync(h->p_z.p, zout, zb, hipMemcpyDeviceToHost, h->stream));  // z_s before guidance
    HIPCHECK(h, hipEventRecord(h->cb_event, h->stream));
    const double t1 = dbg_cb ? now() : 0;
    HIPCHECK(h, hipEventSynchronize(h->cb_event));
    const double t2 = dbg_cb ? now() : 0;
    std::memset(dT, 0, pb);
    if (target_grad_z) {
      float* dz = h->p_dz.as<float>();
      std::memset(dz, 0, zb);
      target_grad_z(user, B, N, D, K, h->p_z.as<float>(), pred, (float)(s + 1) / (float)T, dT, dz);
      // energy = scale * sum_b T (en_diffusion.py:899-903); the reference asserts that the x part of the gradient is zero on
      // masked nodes (remove_mean_with_mask, utils.py:33-44): the direct term is masked here
      for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
          const float m = scale * node_mask[(size_t)b * N + n];
          for (int d = 0; d < D; ++d) dz[((size_t)b * N + n) * D + d] *= m;
        }
      HIPCHECK(h, hipMemcpyAsync(h->d_dz.p, dz, zb, hipMemcpyHostToDevice, h->stream));
      P.dz_in = h->d_dz.as<float>();
    } else {
      target_grad(user, B, K, pred, (float)(s + 1) / (float)T, dT);
    }
    const double t3 = dbg_cb ? now() : 0;
    HIPCHECK(h, hipMemcpyAsync(h->d_dpred.p, dT, pb, hipMemcpyHostToDevice, h->stream));
    // phase B: reverse pass with the caller's dT/dpred, clip / project / apply, CoG removal
    P.mode = gn ? MODE_GUIDE : MODE_SAMPLE;
    P.do_init = 0;
    P.split = 2;
    P.z_in = zout;
    P.z_out = zin;
    rc = launch(h, P, gn ? 0 : h->HPE, h->HPP, 0);
    if (rc) return rc;
    if (dbg_cb) {
      t_enq += (t1 - t0) + (now() - t3);
      t_wait += t2 - t1;
      t_user += t3 - t2;
    }
  }
  if (dbg_cb)
    fprintf(stderr, "[callback] per step: enqueue %.1f us, wait for pred %.1f us, caller's function %.1f us (%d steps)\n",
            1e6 * t_enq / T, 1e6 * t_wait / T, 1e6 * t_user / T, T);
  // decode pass
  P.mode = MODE_SAMPLE;
  P.split = 0;
  P.s_hi = -1;
  P.s_lo = 0;
  P.do_init = 0;
  P.do_decode = 1;
  P.z_in = zin;
  P.z_out = zout;
  rc = launch(h, P, h->HPE, gn ? 0 : h->HPP, 0);
  if (rc) return rc;
  if (z0_out) HIPCHECK(h, hipMemcpyAsync(z0_out, zout, zb, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipMemcpyAsync(x_out, h->d_x.p, sizeof(float) * B * N * 3, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipMemcpyAsync(onehot_out, h->d_h.p, sizeof(float) * B * N * P.F, hipMemcpyDeviceToHost, h->stream));
  int nanc = 0;
  HIPCHECK(h, hipMemcpyAsync(&nanc, h->d_nan.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  finish_sample(B, N, node_mask, x_out, nanc, diag);
  return GAUDI_OK;
}

extern "C" {

int gaudi_sample_cb(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                    int64_t sample_offset, const float* noise, float std, gaudi_target_cb target_grad, void* user,
                    float scale, float* x_out, float* onehot_out, float* z0_out, gaudi_diag* diag) {
  if (!target_grad) return GAUDI_E_INVALID;
  return sample_cb_impl(h, B, N, node_mask, edge_mask, seed, sample_offset, noise, std, target_grad, nullptr, user, scale, x_out,
                        onehot_out, z0_out, diag);
}

int gaudi_sample_cbz(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                     int64_t sample_offset, const float* noise, float std, gaudi_target_cbz target_grad, void* user,
                     float scale, float* x_out, float* onehot_out, float* z0_out, gaudi_diag* diag) {
  if (!target_grad) return GAUDI_E_INVALID;
  return sample_cb_impl(h, B, N, node_mask, edge_mask, seed, sample_offset, noise, std, nullptr, target_grad, user, scale, x_out,
                        onehot_out, z0_out, diag);
}

int gaudi_sample_chain(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                       int64_t sample_offset, const float* noise, float std, int keep_frames, float* chain_out) {
  if (!h || !node_mask || !edge_mask || !chain_out) return GAUDI_E_INVALID;
  if (!h->has_edm) return fail(h, GAUDI_E_STATE, "EDM weights not loaded");
  const int T = h->ecfg.diffusion_steps, F = h->ecfg.in_node_nf, D = 3 + F;
  if (keep_frames < 1 || keep_frames > T) return fail(h, GAUDI_E_INVALID, "keep_frames must be in 1..T");
  std::vector<float> x((size_t)B * N * 3), oh((size_t)B * N * F);
  int rc = run_chain(h, B, N, node_mask, edge_mask, nullptr, true, T - 1, 0, true, noise, 0, T + 2, seed, sample_offset,
                     std, nullptr, 0.f, nullptr, x.data(), oh.data(), nullptr, chain_out, keep_frames);
  if (rc) return rc;
  // chain[0] = cat[x, h_categorical] (en_diffusion.py:1168-1169)
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      float* dst = chain_out + ((size_t)b * N + n) * D;
      for (int d = 0; d < 3; ++d) dst[d] = x[((size_t)b * N + n) * 3 + d];
      for (int k = 0; k < F; ++k) dst[3 + k] = oh[((size_t)b * N + n) * F + k];
    }
  return GAUDI_OK;
}

__global__ void philox_kernel(unsigned long long seed, long long sample_offset, int B, int n_elem, int draw0, int n_draws,
                              float* out) {
  const long long total = (long long)n_draws * B * n_elem;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i % n_elem);
    const int b = (int)((i / n_elem) % B);
    const int dr = (int)(i / ((long long)n_elem * B));
    const f4 v = philox_normal4(seed, (uint64_t)(sample_offset + b), (uint32_t)(draw0 + dr), (uint32_t)(e >> 2));
    out[i] = v[e & 3];
  }
}

int gaudi_philox_normal(gaudi_handle* h, uint64_t seed, int64_t sample_offset, int B, int n_elem, int draw0, int n_draws,
                        float* out) {
  if (!h || !out || B <= 0 || n_elem <= 0 || n_draws <= 0) return GAUDI_E_INVALID;
  HIPCHECK(h, hipSetDevice(h->device));
  const size_t bytes = sizeof(float) * (size_t)B * n_elem * n_draws;
  HIPCHECK(h, h->d_noise.reserve(bytes));
  hipLaunchKernelGGL(philox_kernel, dim3(256), dim3(256), 0, h->stream, seed, sample_offset, B, n_elem, draw0, n_draws,
                     h->d_noise.as<float>());
  HIPCHECK(h, hipGetLastError());
  HIPCHECK(h, hipMemcpyAsync(out, h->d_noise.p, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  return GAUDI_OK;
}

// ---- device-free entry points: the host logic of the library, callable (and tested) without a GPU
int gaudi_host_schedule(int T, float noise_power, float noise_precision, float* gamma_out, float* coef_out) {
  if (T < 1 || !gamma_out || !(noise_power >= 0.f)) return GAUDI_E_INVALID;
  const std::vector<float> g = make_gamma(T, noise_power, noise_precision);
  std::memcpy(gamma_out, g.data(), sizeof(float) * g.size());
  if (coef_out) {
    std::vector<float> c;
    make_coef(g, T, c);
    std::memcpy(coef_out, c.data(), sizeof(float) * c.size());
  }
  return GAUDI_OK;
}

int gaudi_host_graph_meta(int B, int N, const float* node_mask, const float* edge_mask, int32_t* ew_out,
                          int32_t* order_out, int32_t* npairs_out, uint32_t* seg_out, uint32_t* edges_out,
                          float* emask_out, int32_t edges_capacity, int32_t* ncols_out) {
  if (B <= 0 || N <= 0 || !edge_mask || !ew_out) return GAUDI_E_INVALID;
  Meta M;
  std::string err;
  const int rc = build_meta(B, N, node_mask, edge_mask, M, err);
  if (rc) return rc;
  *ew_out = M.EW;
  if (order_out) std::memcpy(order_out, M.order.data(), sizeof(int) * B);
  if (npairs_out) std::memcpy(npairs_out, M.npairs.data(), sizeof(int) * B * kWaves);
  if (seg_out) std::memcpy(seg_out, M.seg.data(), sizeof(uint32_t) * B * N);
  if (ncols_out) std::memcpy(ncols_out, M.ncols.data(), sizeof(int) * B);
  if (edges_out || emask_out) {
    if ((size_t)edges_capacity < M.edges.size()) return GAUDI_E_CAPACITY;
    if (edges_out) std::memcpy(edges_out, M.edges.data(), sizeof(uint32_t) * M.edges.size());
    if (emask_out) std::memcpy(emask_out, M.emask.data(), sizeof(float) * M.emask.size());
  }
  return GAUDI_OK;
}

int gaudi_host_graph_meta8(int B, int N, const float* node_mask, const float* edge_mask, int32_t* slots_out,
                           int32_t* order_out, int32_t* ntiles_out, uint32_t* seg_out, uint32_t* edges_out, float* emask_out,
                           uint16_t* soff_out, uint16_t* sidx_out, int32_t edges_capacity, int32_t* ncols_out) {
  if (B <= 0 || N <= 0 || !edge_mask || !slots_out) return GAUDI_E_INVALID;
  Meta8 M;
  std::string err;
  const int rc = build_meta8(B, N, node_mask, edge_mask, M, err);
  if (rc) return rc;
  *slots_out = M.S;
  if (order_out) std::memcpy(order_out, M.order.data(), sizeof(int) * B);
  if (ntiles_out) std::memcpy(ntiles_out, M.ntiles.data(), sizeof(int) * B);
  if (seg_out) std::memcpy(seg_out, M.seg.data(), sizeof(uint32_t) * B * N);
  if (ncols_out) std::memcpy(ncols_out, M.ncols.data(), sizeof(int) * B);
  if (soff_out) std::memcpy(soff_out, M.soff.data(), sizeof(uint16_t) * M.soff.size());
  if (edges_out || emask_out || sidx_out) {
    if ((size_t)edges_capacity < M.edges.size()) return GAUDI_E_CAPACITY;
    if (edges_out) std::memcpy(edges_out, M.edges.data(), sizeof(uint32_t) * M.edges.size());
    if (emask_out) std::memcpy(emask_out, M.emask.data(), sizeof(float) * M.emask.size());
    if (sidx_out) std::memcpy(sidx_out, M.sidx.data(), sizeof(uint16_t) * M.sidx.size());
  }
  return GAUDI_OK;
}

int gaudi_host_pack_plan(int B, int N, const float* node_mask, const float* edge_mask, int32_t* groups_out, int32_t* group_of_out,
                         int32_t* ntiles_out, int32_t* ncols_out) {
  if (B <= 0 || N <= 0 || !node_mask || !edge_mask || !groups_out) return GAUDI_E_INVALID;
  Meta8 M;
  std::string err;
  int rc = build_meta8(B, N, node_mask, edge_mask, M, err);
  if (rc) return rc;
  Pack pk;
  pack_groups(B, N, node_mask, edge_mask, M, pk);
  Meta8 M2;
  rc = build_meta8(pk.G, N, pk.umask.data(), pk.uemask.data(), M2, err, M.S, pk.align.data());
  if (rc) return rc;
  *groups_out = pk.G;
  for (int g = 0; g < pk.G; ++g) {
    if (ntiles_out) ntiles_out[g] = M2.ntiles[g];
    if (ncols_out) ncols_out[g] = M2.ncols[g];
    if (group_of_out)
      for (int k = 0; k < pk.ncomp[g]; ++k) group_of_out[pk.compmol[(size_t)g * kMaxComp + k]] = g;
  }
  return GAUDI_OK;
}

int gaudi_host_pack_plan_wide(int B, int N, int node_slots, int tiles, const float* node_mask, const float* edge_mask,
                              int32_t* groups_out, int32_t* group_of_out, int32_t* ntiles_out, int32_t* ncols_out) {
  if (B <= 0 || N <= 0 || node_slots < N || node_slots > 255 || tiles < 1 || !node_mask || !edge_mask || !groups_out) return GAUDI_E_INVALID;
  Meta8 M;
  std::string err;
  int rc = build_meta8(B, N, node_mask, edge_mask, M, err);
  if (rc) return rc;
  Pack pk;
  pack_groups(B, N, node_mask, edge_mask, M, pk, node_slots, tiles);
  Meta8 M2;
  rc = build_meta8(pk.G, node_slots, pk.umask.data(), pk.uemask.data(), M2, err, 0, pk.align.data());
  if (rc) return rc;
  *groups_out = pk.G;
  for (int g = 0; g < pk.G; ++g) {
    if (ntiles_out) ntiles_out[g] = M2.ntiles[g];
    if (ncols_out) ncols_out[g] = M2.ncols[g];
    if (group_of_out)
      for (int k = 0; k < pk.ncomp[g]; ++k) group_of_out[pk.compmol[(size_t)g * kMaxComp + k]] = g;
  }
  return GAUDI_OK;
}

int gaudi_kernel_variant(const gaudi_handle* h, int32_t* configured, int32_t* last_call) {
  if (!h) return GAUDI_E_INVALID;
  if (configured) *configured = h->variant;
  if (last_call) *last_call = h->run_variant;
  return GAUDI_OK;
}

int gaudi_last_workgroups(const gaudi_handle* h, int32_t* workgroups, int32_t* node_slots) {
  if (!h || !workgroups) return GAUDI_E_INVALID;
  *workgroups = h->run_groups;
  if (node_slots) *node_slots = h->run_nslots;
  return GAUDI_OK;
}

int gaudi_node_buffers(const gaudi_handle* h, int32_t* last_call) {
  if (!h || !last_call) return GAUDI_E_INVALID;
  // 0 resident, 1 global scratch, 2 global scratch with P / Q in LDS (8-wave kernels), 3 resident except the predictor's fifth buffer
  *last_call = h->run_gn ? 1 : h->run_pg ? 3 : h->run_gn8;
  return GAUDI_OK;
}

int gaudi_edge_math(const gaudi_handle* h, int32_t* configured, int32_t* last_call) {
  if (!h) return GAUDI_E_INVALID;
  if (configured) *configured = (h->variant == 8 && h->split) ? 1 : 0;
  if (last_call) *last_call = h->run_split;
  return GAUDI_OK;
}

int gaudi_set_plan_hint(gaudi_handle* h, int32_t min_slots, int32_t force_waves) {
  if (!h || min_slots < 0 || (force_waves != 0 && force_waves != 4)) return GAUDI_E_INVALID;
  h->plan_min_slots = min_slots;
  h->plan_force_waves = force_waves;
  return GAUDI_OK;
}

int gaudi_host_pack_matrix(int H, int ldw, int col0, int HP, int transpose, const float* W, float* packed_out) {
  if (H < 1 || HP < H || HP % 16 || !W || !packed_out) return GAUDI_E_INVALID;
  std::memset(packed_out, 0, sizeof(float) * (size_t)HP * HP);
  pack_matrix(PackMode{}, packed_out, W, H, ldw, col0, HP, transpose != 0);
  return GAUDI_OK;
}

int gaudi_host_pack_matrix_split(int H, int ldw, int col0, int HP, int transpose, int ktail, float scale, const float* W,
                                 float* packed_out) {
  if (H < 1 || HP < H || HP % 16 || !W || !packed_out || !(scale > 0.f)) return GAUDI_E_INVALID;
  const int T = HP / 16;
  std::memset(packed_out, 0, sizeof(float) * (size_t)((T + 1) / 2) * T * w8::kPieces * 256);
  PackMode pm;
  pm.lane_linear = true;
  pm.ktail = ktail != 0 && has_ktail(H, HP);
  pm.hscale = scale;
  pack_matrix_split(pm, packed_out, W, H, ldw, col0, HP, transpose != 0);
  return GAUDI_OK;
}

int gaudi_host_pack_matrix_f16(int H, int ldw, int col0, int HP, int transpose, float scale, const float* W, float* packed_out) {
  if (H < 1 || HP < H || HP % 16 || !W || !packed_out || !(scale > 0.f)) return GAUDI_E_INVALID;
  std::memset(packed_out, 0, sizeof(float) * (size_t)HP * HP);
  pack_matrix_f16(packed_out, W, H, ldw, col0, HP, transpose != 0, scale);
  return GAUDI_OK;
}

int gaudi_host_weight_scale(int n, const float* const* blocks, const int32_t* rows, const int32_t* cols, const int32_t* ldw,
                            float* scale_out) {
  if (n < 1 || !blocks || !rows || !cols || !ldw || !scale_out) return GAUDI_E_INVALID;
  NodeScale ns;
  for (int i = 0; i < n; ++i) ns.see(blocks[i], rows[i], ldw[i], 0, cols[i], true);  // (as edge-level matrices: the stricter floor)
  *scale_out = ns.scale();
  return GAUDI_OK;
}

int gaudi_host_node_operand_offset(int column_tiles, int chunk, int tile, int c, int g, int32_t* float_offset_out) {
  if (column_tiles < 1 || column_tiles > 3 || chunk < 0 || tile < 0 || tile >= column_tiles || c < 0 || c > 15 || g < 0 || g > 3 || !float_offset_out)
    return GAUDI_E_INVALID;
  *float_offset_out = chunk * w8::nh_chunk_stride(column_tiles) + tile * 512 + w8::nh_bpos(c, g);
  return GAUDI_OK;
}

int gaudi_profile_reset(gaudi_handle* h, int enable) {
  if (!h) return GAUDI_E_INVALID;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->prof_log.reset(false);
  h->stab_log.reset(false);
  h->prof_steps = 0;
  h->prof = enable != 0;
  return GAUDI_OK;
}

int gaudi_profile_get(gaudi_handle* h, int32_t* n_launches, double* total_ms, int64_t* steps_done) {
  if (!h) return GAUDI_E_INVALID;
  HIPCHECK(h, hipStreamSynchronize(h->stream));
  HIPCHECK(h, h->prof_log.fold(0));
  if (n_launches) *n_launches = (int32_t)h->prof_log.n;
  if (total_ms) *total_ms = h->prof_log.ms;
  if (steps_done) *steps_done = h->prof_steps;
  return GAUDI_OK;
}

int gaudi_set_fix_noise(gaudi_handle* h, int enable, int64_t key_sample) {
  if (!h) return GAUDI_E_INVALID;
  h->fix_noise = enable != 0;
  h->fix_key = key_sample;
  return GAUDI_OK;
}

int gaudi_set_steps_per_launch(gaudi_handle* h, int steps) {
  if (!h || steps < 1) return GAUDI_E_INVALID;
  h->steps_per_launch = steps;
  return GAUDI_OK;
}

}  // extern "C"

#include "pred_host.inc"
#include "stability.inc"
