/*
 * gaudi_hip.h -- C ABI of libgaudi_hip.so: MI355X (gfx950) guided-diffusion sampler for GaUDI.
 *
 * The reference (tomer196/GaUDI) has no FFI/plugin layer: its boundary for this path is a set
 * of Python call signatures plus an on-disk checkpoint.  This header is
 * therefore the boundary a maintainer would bind with ctypes; each entry
 * point names the reference function it replaces.  Conventions:
 *   - every function returns 0 on success or a negative GAUDI_E_* code; nothing throws across
 *     the ABI; gaudi_last_error(h) returns a human-readable message for the last failure;
 *   - all pointers are HOST pointers to caller-owned, contiguous fp32/int buffers; the handle
 *     owns all device memory and one HIP stream; one handle per device, not thread-safe;
 *   - tensors use the reference's layouts: z/eps/grad [B,N,3+F] (x first), node_mask [B,N],
 *     edge_mask [B,N,N] (row i = receiving node, as edm/egnn/models.py:154-175).
 */
#ifndef GAUDI_HIP_H
#define GAUDI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAUDI_OK 0
#define GAUDI_E_INVALID (-1)   /* bad argument / unsupported configuration      */
#define GAUDI_E_HIP (-2)       /* HIP runtime error                              */
#define GAUDI_E_STATE (-3)     /* e.g. sampling before weights are loaded        */
#define GAUDI_E_MISSING (-4)   /* a required checkpoint tensor was not supplied  */
#define GAUDI_E_CAPACITY (-5)  /* problem does not fit the kernel's LDS budget   */

typedef struct gaudi_handle gaudi_handle;

/* Architecture of EnVariationalDiffusion(EGNN_dynamics): models_edm.py:67-96 / utils/args_edm.py. */
typedef struct {
  int32_t in_node_nf;          /* F: dataset.num_node_features (cata 1, hetro 12)            */
  int32_t hidden_nf;           /* args.nf                                                      */
  int32_t n_layers;            /* args.n_layers (EquivariantBlocks)                            */
  int32_t inv_sublayers;       /* args.inv_sublayers (GCLs per block)                          */
  int32_t attention;           /* args.attention                                               */
  int32_t tanh;                /* args.tanh                                                    */
  float coords_range;          /* args.coords_range (NOT divided by n_layers, egnn_new.py:290) */
  float norm_constant;         /* args.norm_constant                                           */
  float normalization_factor;  /* args.normalization_factor of aggregation_method "sum"; 0 = aggregation_method "mean": divide by the
                                  number of edges of the dense list per row, masked ones included = the call's padded node
                                  count N (unsorted_segment_sum, edm/egnn/egnn_new.py:403-421)   */
  int32_t diffusion_steps;     /* args.diffusion_steps (T)                                     */
  float noise_power;           /* p of "polynomial_<p>" (en_diffusion.py:47-61); 0 = the "cosine" schedule (:64-81,196-197) */
  float noise_precision;       /* args.diffusion_noise_precision                               */
  float norm_values[3];        /* args.normalize_factors                                       */
  int32_t sin_embedding;       /* args.sin_embedding (egnn_new.py:269-273,378-391): the first Linear of every edge / coordinate MLP
                                  takes 2 x 12 sinusoids of sqrt(r), sqrt(d0) instead of (r, d0).  Such a denoiser runs on the
                                  4-wave kernels (about 0.4 x the 8-wave rate; fused with the predictor at the tiny and the default
                                  width pairs, two launches per guided step elsewhere)   (ABI 7)   */
} gaudi_edm_config;

/* Architecture of EGNN_predictor: cond_prediction/train_cond_predictor.py:183-196. */
typedef struct {
  int32_t in_nf;        /* F                                                      */
  int32_t out_nf;       /* K = dataset.num_targets                                */
  int32_t hidden_nf;    /* args.nf                                                */
  int32_t n_layers;     /* args.n_layers                                          */
  int32_t attention;
  int32_t tanh;
  float coords_range;   /* divided by n_layers inside (egnn_predictor/models.py:515) */
} gaudi_pred_config;

/* Per-call diagnostics replacing the reference's per-step asserts
 * (edm/equivariant_diffusion/utils.py:52-65, sampling_edm.py:167-168,222-223). */
typedef struct {
  float max_masked_leak;   /* max |x * (1-node_mask)|                     */
  float max_cog_rel;       /* max |sum_n x| / (max|x| + 1e-10)            */
  float max_cog_abs;       /* max |sum_n x| before the final re-projection */
  int32_t nan_count;       /* NaNs scrubbed inside the chain              */
  int32_t reprojected;     /* 1 if the 5e-2 CoG re-projection fired (en_diffusion.py:1000-1006) */
} gaudi_diag;

int gaudi_create(int device, gaudi_handle** out);
void gaudi_destroy(gaudi_handle* h);
const char* gaudi_last_error(const gaudi_handle* h);
/* The last NON-fatal condition of the handle a caller should know about ("" if none): e.g. a weight set the fp16-pair images
 * cannot carry (an infinite weight, a matrix far below the others), whose calls therefore run the fp32-instruction kernels at
 * about 0.55 x the speed.  gaudi_amd.engine turns it into a Python warning at load time and into diag["edge_math_fallback"]. */
const char* gaudi_last_warning(const gaudi_handle* h);
/* Bumped whenever an exported signature or a config struct changes (round 6: 6; 7: gaudi_edm_config.sin_embedding appended).  gaudi_amd/_lib.py refuses to bind the host-side packers of a
 * diagnostic library (GAUDI_LIB) whose version differs: round 5 inserted an argument into gaudi_host_pack_matrix_split. */
#define GAUDI_ABI_VERSION 7
int gaudi_abi_version(void);

/* Load a state dict (reference key names WITHOUT the "module." prefix).
 * names[i] is the key, tensors[i] its fp32 data, numel[i] its element count.  Unknown keys are
 * ignored ("buffer", "gamma.gamma": the schedule is rebuilt from the config exactly as
 * PredefinedNoiseSchedule does); missing ones fail with GAUDI_E_MISSING.
 * Replaces models_edm.get_model / load_state_dict (models_edm.py:61-104). */
int gaudi_load_edm(gaudi_handle* h, const gaudi_edm_config* cfg, int n, const char* const* names,
                   const float* const* tensors, const int64_t* numel);
/* Replaces get_cond_predictor_model (cond_prediction/train_cond_predictor.py:183-203). */
int gaudi_load_predictor(gaudi_handle* h, const gaudi_pred_config* cfg, int n, const char* const* names,
                         const float* const* tensors, const int64_t* numel);

/* gamma table [T+1] as PredefinedNoiseSchedule builds it (en_diffusion.py:191-218). */
int gaudi_get_gamma(gaudi_handle* h, float* gamma_out /* [T+1] */);
/* Per-step scalars (en_diffusion.py:433-457, 811-821): coef_out[s] = {alpha_t|s, sigma2_t|s/alpha_t|s/sigma_t,
 * sigma_t|s*sigma_s/sigma_t, t=(s+1)/T} for s in [0,T). */
int gaudi_get_step_coefficients(gaudi_handle* h, float* coef_out /* [T][4] */);

/* eps_hat = EGNN_dynamics._forward(t, z, node_mask, edge_mask)  (edm/egnn/models.py:83-152). */
int gaudi_phi(gaudi_handle* h, int B, int N, const float* z, const float* t /* [B] */,
              const float* node_mask, const float* edge_mask, float* eps_out);

/* pred = EGNN_predictor.forward(z, node_mask, edge_mask, t)  (edm/egnn_predictor/models.py:433-457). */
int gaudi_predictor_fwd(gaudi_handle* h, int B, int N, const float* z, const float* t /* [B] */,
                        const float* node_mask, const float* edge_mask, float* pred_out /* [B,K] */);
/* grad = d(sum_b dpred[b].pred[b])/dz: the hand-written reverse pass replacing
 * torch.autograd.grad at en_diffusion.py:900-903. */
int gaudi_predictor_grad(gaudi_handle* h, int B, int N, const float* z, const float* t, const float* node_mask,
                         const float* edge_mask, const float* dpred /* [B,K] */, float* pred_out /* [B,K] or NULL */,
                         float* grad_out /* [B,N,3+F] */);

/* Forward noising + predictor evaluation at noise level t: sample_edm_t (cond_prediction/train_cond_predictor.py:47-61)
 * followed by the predictor forward of compute_loss (:64-81) / eval_cond_predictor.val_epoch (eval_cond_predictor.py:34-60).
 * x [B,N,3] (un-normalised, masked, mean-free) and onehot [B,N,F] are normalised as EnVariationalDiffusion.normalize does,
 * z_t = alpha_t * xh + sigma_t * eps with gamma looked up at t_int[b] in 0..T and eps = the combined position/feature noise
 * (injected raw draws `noise` [B,N,3+F], or Philox draw 0 of (seed, sample_offset + b) when NULL);
 * pred = predictor(z_t, t_int/T).  Either output may be NULL. */
int gaudi_predict_noised(gaudi_handle* h, int B, int N, const float* x, const float* onehot, const int32_t* t_int,
                         const float* node_mask, const float* edge_mask, uint64_t seed, int64_t sample_offset,
                         const float* noise, float* zt_out /* [B,N,3+F] or NULL */, float* pred_out /* [B,K] or NULL */);

/* One teacher-forced reverse step z_t -> z_s with s = s_idx/T, t = (s_idx+1)/T:
 * sample_p_zs_given_zt (en_diffusion.py:807-852) when target_w == NULL, else
 * sample_p_zs_given_zt_guidance (:854-935) for the target  T(pred) = target_w . pred  scaled by `scale`.
 * eps_raw [B,N,3+F] are the raw N(0,1) draws (x part first). */
int gaudi_step(gaudi_handle* h, int B, int N, int s_idx, const float* z_t, const float* node_mask,
               const float* edge_mask, const float* eps_raw, const float* target_w /* [K] or NULL */,
               float scale, float* zs_out);

/* x, one_hot = sample_p_xh_given_z0(z0)  (en_diffusion.py:533-560 + unnormalize :406-415). */
int gaudi_decode(gaudi_handle* h, int B, int N, const float* z0, const float* node_mask, const float* edge_mask,
                 const float* eps_raw, float* x_out /* [B,N,3] */, float* onehot_out /* [B,N,F] */);

/* Whole chain: EnVariationalDiffusion.sample (en_diffusion.py:958-1008) when target_w == NULL, else
 * .sample_guidance (:1010-1067).  noise: NULL -> on-device Philox4x32-10 keyed by
 * (seed, sample_offset + b, draw, element) so results do not depend on how samples are sharded;
 * otherwise injected raw N(0,1) draws [T+2,B,N,3+F] (draw 0 -> z_T, 1+k -> k-th step, T+1 -> decode).
 * Shards of one logical batch must all use the batch-wide padded N (sample_guidance pads to the
 * batch max, sampling_edm.py:177): pad the masks to that N before calling. */
int gaudi_sample(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                 int64_t sample_offset, const float* noise, float std, const float* target_w /* [K] or NULL */,
                 float scale, float* x_out /* [B,N,3] */, float* onehot_out /* [B,N,F] */,
                 float* z0_out /* [B,N,3+F] or NULL */, gaudi_diag* diag /* or NULL */);

/* sample_guidance for an ARBITRARY differentiable target T(pred, t) (the reference accepts any closure over the
 * predictor, generation_guidance.py:187-205).  Each reverse step runs in two launches: (A) denoise + predictor forward,
 * (B) predictor reverse pass + guidance update; in between, target_grad receives pred [B,K] and t and must write
 * dT/dpred [B,K] (the energy is scale * sum_b T(pred_b), en_diffusion.py:899).  Both networks stay on the device;
 * only the K-vector per molecule crosses the boundary (pinned host buffers, one event wait per step).  With target_grad
 * returning a constant w this equals gaudi_sample(target_w = w) bit for bit.  Molecules beyond the LDS limit (the V4G
 * kernels: node buffers in global memory) run phase A as two launches -- the reference has no size cap,
 * sampling_edm.py:172-209 -- so a step there is three launches. */
typedef void (*gaudi_target_cb)(void* user, int B, int K, const float* pred, float t, float* dT_dpred_out);
int gaudi_sample_cb(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                    int64_t sample_offset, const float* noise, float std, gaudi_target_cb target_grad, void* user,
                    float scale, float* x_out, float* onehot_out, float* z0_out, gaudi_diag* diag);

/* The same for a target that ALSO depends on z outside the predictor -- the reference differentiates any function of z_s
 * (autograd at en_diffusion.py:899-903).  Between the two phases the callback receives z_s [B,N,D] (D = 3 + F, before the
 * guidance update) beside pred and t and writes dT/dpred [B,K] and the DIRECT part dT/dz [B,N,D] (pred held fixed); the
 * library scales it, masks it with node_mask (the reference asserts that the coordinate gradient of masked nodes is zero,
 * utils.py:33-44) and adds it to the reverse pass's gradient before the clip (en_diffusion.py:905-909). */
typedef void (*gaudi_target_cbz)(void* user, int B, int N, int D, int K, const float* z_s, const float* pred, float t,
                                 float* dT_dpred_out, float* dT_dz_out);
int gaudi_sample_cbz(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                     int64_t sample_offset, const float* noise, float std, gaudi_target_cbz target_grad, void* user,
                     float scale, float* x_out, float* onehot_out, float* z0_out, gaudi_diag* diag);

/* EnVariationalDiffusion.sample_chain (en_diffusion.py:1118-1174): the unguided chain with `keep_frames`
 * intermediate states: chain_out [keep_frames,B,N,3+F], frame (s*keep_frames)//T = unnormalize_z(z_s) of the last
 * step s mapping to it, frame 0 = the final [x | one_hot].  (The reference returns the same data viewed as
 * [keep_frames*B, N, 3+F].) */
int gaudi_sample_chain(gaudi_handle* h, int B, int N, const float* node_mask, const float* edge_mask, uint64_t seed,
                       int64_t sample_offset, const float* noise, float std, int keep_frames, float* chain_out);

/* ---- Graph-of-rings stability check: the step that follows sampling (eval_validity.py:49, sampling_edm.py:96). ----
 * Geometry tables of one dataset, indexed by ring-type index (= class index of the one-hot node features,
 * data/aromatic_dataloader.py:31-35).  Raw values of utils/helpers.py:11-157 in double; the library applies `tol`
 * as the reference does ((1 - tol) on lower, (1 + tol) on upper bounds, then fp32). */
typedef struct {
  int32_t n_types;          /* len(RINGS_LIST[dataset]) <= 16                                               */
  int32_t orientation;      /* 1 if dataset != "cata": nodes [n/2, n) are orientation nodes of type n_types-1 */
  int32_t check_dihedrals;  /* 0 for "hetro" (check_angels4 returns True, analyze/analyze.py:40)              */
  double tol;               /* 0.1 in every reference call                                                    */
  double min_dist;          /* min over ring_distances[dataset] of the lower bound                            */
  double dist_lo[16][16], dist_hi[16][16]; /* ring_distances window of the type pair; hi == 0: never bonded   */
  int32_t a3_count[16];     /* number of 3-ring angle windows of the centre type (<= 4)                       */
  double a3_lo[16][4], a3_hi[16][4];       /* angels3_dict[dataset][type]                                     */
  double a4_0, a4_180;      /* angels4_dict[dataset]                                                          */
} gaudi_ring_tables;

typedef struct {
  int32_t n_rings;       /* rings tested (orientation nodes excluded)              */
  int32_t n_edges;       /* bonded ring pairs                                      */
  int32_t n_triplets;    /* de-duplicated 3-ring paths the reference enumerates    */
  int32_t n_nan_angles;  /* 3-ring angles that came out NaN (acos of 1 + 1 ulp)    */
  float a3_min, a3_max;  /* range of the 3-ring angles (degrees; +-inf when none)  */
  float a4_min, a4_max;  /* range of the 4-ring dihedrals tested                   */
} gaudi_stability_aux;

/* check_stability (analyze/analyze.py:50-100) for B molecules at once: x [B,N,3] and ring_type [B,N] hold each
 * molecule's n_nodes[b] valid nodes first (the reference compacts with node_mask before the call).
 * flags_out [B,5] = {orientation_nodes, dist_stable, connected, angels3, angels4}; a molecule is stable when all
 * five are 1.  Optional outputs: dist_out / adj_out [B,N,N] = positions2adj (utils/helpers.py:167-190) of the ring
 * block, zero elsewhere; aux_out [B]. */
int gaudi_check_stability(gaudi_handle* h, const gaudi_ring_tables* tables, int B, int N, const float* x,
                          const int32_t* ring_type, const int32_t* n_nodes, uint8_t* flags_out,
                          float* dist_out /* or NULL */, float* adj_out /* or NULL */,
                          gaudi_stability_aux* aux_out /* or NULL */);
/* Number of stability-kernel launches since gaudi_profile_reset(h, 1) and their summed duration (HIP events). */
int gaudi_stability_profile_get(gaudi_handle* h, int32_t* n_launches, double* total_ms);

/* Device Philox stream used when noise == NULL, exposed for tests: out[draw][b][e], e < n_elem. */
int gaudi_philox_normal(gaudi_handle* h, uint64_t seed, int64_t sample_offset, int B, int n_elem, int draw0,
                        int n_draws, float* out);

/* ---- Device-free host logic (no handle, no GPU): exposed so the CPU test suite can check it. ----
 * gamma [T+1] and (optionally) the per-step table [T][4] exactly as gaudi_load_edm builds them. */
int gaudi_host_schedule(int T, float noise_power /* 0 = cosine */, float noise_precision, float* gamma_out, float* coef_out);
/* The live-edge metadata gaudi_* calls derive from (node_mask, edge_mask): per-wave capacity EW (multiple of 32),
 * launch order [B] (heaviest first), 32-edge passes per wave [B][4], per-node segment word [B][N]
 * (wave<<30 | start<<15 | len), the padded per-wave edge lists [B][4][EW] (i | j<<8) with their mask values, and
 * ncols [B] = 1 + the last node that is live or touches a live edge (node-level GEMMs stop there).  Live = mask != 0 and
 * not both endpoints masked (the reference's unmasked "padded ring <-> orientation node" identity edges reach no live
 * node).  edges_capacity = number of entries edges_out / emask_out can hold (B*4*EW needed). */
int gaudi_host_graph_meta(int B, int N, const float* node_mask, const float* edge_mask, int32_t* ew_out,
                          int32_t* order_out, int32_t* npairs_out, uint32_t* seg_out, uint32_t* edges_out,
                          float* emask_out, int32_t edges_capacity, int32_t* ncols_out /* [B] or NULL */);
/* The metadata of the 8-wave kernels (default): per molecule ONE flat slot list = its live edges sorted by receiving node
 * (then sending node), padded to whole 16-slot tiles (tile tau -> wave tau & 7 of the workgroup, round tau >> 3).
 * slots_out = slot capacity of the batch (multiple of 16); ntiles [B]; seg [B][N] = first_slot << 16 | run length of the
 * edges RECEIVED by a node; edge words [B][slots] = i | j << 8 | run_start_column << 16 | run_end << 20 | partial << 21
 * (a node's run touches at most two tiles; partial = 1 for its second tile) with their mask values; soff [B][N+1] / sidx
 * [B][slots] = CSR lists of the slots whose SENDING node is n (ascending), which the reverse pass sums over. */
int gaudi_host_graph_meta8(int B, int N, const float* node_mask, const float* edge_mask, int32_t* slots_out,
                           int32_t* order_out, int32_t* ntiles_out, uint32_t* seg_out, uint32_t* edges_out, float* emask_out,
                           uint16_t* soff_out, uint16_t* sidx_out, int32_t edges_capacity, int32_t* ncols_out);
/* Kernel family of a handle: 8 = eight waves per molecule, two per SIMD (default); 4 = four waves, one per SIMD
 * (environment GAUDI_WAVES=4 at gaudi_create, and the per-call fallback of an 8-wave handle for graphs beyond the 8-wave
 * kernels' limits: more than 128 edge slots with guidance, a node with more than 32 live edges, LDS).  last_call = what the
 * most recent call ran on. */
int gaudi_kernel_variant(const gaudi_handle* h, int32_t* configured, int32_t* last_call);
/* Arithmetic of the GEMMs (edge level: the W2 / Wc1 contractions and their transposes, edm/egnn/egnn_new.py:42-47 and
 * edm/egnn_predictor/gcl.py:225-231; since round 5 the node-level Linears too) on the 8-wave kernels: 1 = every fp32 operand
 * as a pair of fp16 pieces behind exact power-of-two scales, three piece products accumulated in fp32 on the fp16 matrix pipe
 * (rounds 2-4: three bf16 pieces, six products; error against float64 at or below the fp32 matrix instruction's:
 * tests/test_gpu_split.py, tests/test_gpu_round5.py); 0 = v_mfma_f32_16x16x4_f32 (environment GAUDI_EDGE_MATH=fp32 at gaudi_create,
 * the 4-wave kernels, and the per-call fallback when the LDS weight ring of the split form does not fit).  last_call: 1 =
 * split with the full ring (a 32-input chunk of all output tiles per trip), 2 = split with the half ring (two trips per
 * chunk: molecules whose node buffers leave less LDS), 0 = fp32 instructions. */
int gaudi_edge_math(const gaudi_handle* h, int32_t* configured, int32_t* last_call);
/* Where the most recent call kept its [N][hidden] node buffers: 0 = LDS (the resident kernels), 1 = a per-workgroup global
 * scratch -- molecules beyond the LDS limit (about 22 graph nodes at the default widths; the reference has no cap,
 * sampling_edm.py:172-209): the V8G kernels (8 waves, round 4) or, for graphs outside the 8-wave kernels' limits and with
 * GAUDI_GN8=0, the V4G kernels (4 waves; gaudi_kernel_variant tells which); 2 = the V8G kernels' hybrid form (round 6): the global
 * scratch holds three of the five buffers, P and Q stay in LDS (taken where that plan fits; GAUDI_GN8_PQ=0 turns it off); 3 = a WIDE
 * group (two molecules per workgroup) on the full weight ring: everything in LDS except one of the predictor's five buffers
 * (round 6; GAUDI_WIDE_FULL=0: the half ring with all five in LDS, as in round 5 -- the same bits). */
int gaudi_node_buffers(const gaudi_handle* h, int32_t* last_call);
/* Workgroups the most recent kernel launch of the handle ran: the molecules of the call (or of its last sub-batch), or the
 * groups they were packed into; node_slots (may be NULL): node slots per workgroup -- the call's N, or more when the launch
 * ran WIDE groups (below).  bench.py prices its roofline with these figures, not with the host-side plan. */
int gaudi_last_workgroups(const gaudi_handle* h, int32_t* workgroups, int32_t* node_slots);
/* Shader clock (MHz) the chip held during the most recent PROFILED sampler launch (gaudi_profile_reset(h, 1)): cycles of the
 * shader clock over ticks of the constant 100 MHz counter, both read by workgroup 0 at kernel entry and at the end of its last
 * step.  0 if no profiled launch has run.  The roofline peaks in bench.py assume the nominal 2.4 GHz; under this kernel's load
 * the chip runs at 1.7-2.0 GHz. */
int gaudi_profile_clock(gaudi_handle* h, double* shader_mhz);
/* Floats of LDS the most recent call gave to a KEPT split copy of h (round 6; 0: none -- no room in 160 KiB, a kernel family
 * without fp16-pair node GEMMs, or GAUDI_KEEP_H=0 at gaudi_create).  With it the node GEMMs that read h (P, Q, the node MLP's first
 * Linear: edm/egnn/egnn_new.py:59-73,119-128, edm/egnn_predictor/gcl.py:240-250) split it into fp16 pairs once per change of h
 * instead of once per GEMM; the copy is a function of h alone, results do not change by a bit. */
int gaudi_last_keep_h(const gaudi_handle* h, int32_t* lds_floats);
/* Per-molecule kernel family (round 6).  gaudi_sample sorts a request whose padded N is beyond the resident kernels' LDS limit
 * into the molecules that fit those kernels on their own (at most as many live nodes as the widest resident group takes, one round
 * of eight edge tiles -- a function of the molecule's own graph and the hidden sizes, so the choice does not depend on the rest of
 * the batch or on how it is sharded) and the rest; the first bucket runs packed on the resident kernels, the second on the V8G
 * kernels, both with noise keyed by the molecule's index in the request.  resident_molecules: how many molecules of the most
 * recent gaudi_sample call ran in the first bucket (0: the call ran one family).  Matches sampling_edm.py:172-209 (no size cap,
 * mixed sizes in one call).  OPT-IN (environment GAUDI_FAMILY_SPLIT=1 at gaudi_create): the two buckets are two launches per 25
 * steps on one stream, each with its own tail of idle CUs, and BASELINE config 4 read literally measures 76.8 mol/s this way
 * against 84.3 in one family (round 6); results are bit-identical per molecule either way of sharding. */
int gaudi_last_family_split(const gaudi_handle* h, int32_t* resident_molecules);
/* The same plan for WIDE groups (device-free): groups of up to node_slots (>= N) node slots and `tiles` edge tiles.  Opt-in
 * (environment GAUDI_PAIRS at gaudi_create: 0 never = default, 1 for batches of at least two molecules per CU, 2 always): a
 * sampling call then gives a workgroup more node slots than a molecule has -- two 11-ring cata molecules, three or four small
 * hetero ones -- and two rounds of eight edge tiles, so that every node-level weight matrix is streamed from L2 once for all of
 * them.  Results do not depend on the grouping, bit for bit (tests/test_gpu_round4.py); the guided path measured 2 % slower
 * this way at 1024 molecules, which is why it is not the default. */
int gaudi_host_pack_plan_wide(int B, int N, int node_slots, int tiles, const float* node_mask, const float* edge_mask,
                              int32_t* groups_out, int32_t* group_of_out, int32_t* ntiles_out, int32_t* ncols_out);
/* How a sampling call of the 8-wave kernels packs a batch (device-free): small molecules share a workgroup as the components
 * of one disjoint graph -- at most 4 molecules, N node slots and 8 edge tiles of 16 slots per group; a molecule keeps its own
 * tiles, node order, noise stream and per-molecule reductions, so its result does not depend on the packing (tested bit for
 * bit; GAUDI_PACK=0 at gaudi_create turns it off).  groups_out = G; group_of_out[B] = group of every molecule; ntiles_out /
 * ncols_out [>= G] = edge tiles and node columns of each group (what bench.py counts the issued matrix instructions from).
 * No reference counterpart (the reference batches dense N x N tensors, sampling_edm.py:172-209). */
int gaudi_host_pack_plan(int B, int N, const float* node_mask, const float* edge_mask, int32_t* groups_out, int32_t* group_of_out,
                         int32_t* ntiles_out, int32_t* ncols_out);
/* Plan hint for shards of a larger logical batch (gaudi_amd/dist.py; no reference counterpart: the reference runs one
 * process).  The kernel family and the edge-GEMM arithmetic of a call are chosen from batch-wide maxima -- the largest
 * number of 16-edge slots of a molecule (gaudi_host_graph_meta8: slots_out) and whether any node has more than 32 live edges
 * (gaudi_host_graph_meta8 returns GAUDI_E_CAPACITY) -- so two shards of one batch could otherwise run different arithmetic
 * and the gathered result would depend on the cut.  min_slots: plan every following call as if its batch held a molecule
 * with that many slots (0 = no hint); force_waves: 4 = run every following call on the 4-wave kernels, 0 = automatic.
 * gaudi_sample applies the same rule by itself to the sub-batches it cuts a large request into. */
int gaudi_set_plan_hint(gaudi_handle* h, int32_t min_slots, int32_t force_waves);
/* Tile packing of a weight block W[o][col0+k] (o,k < H, row stride ldw) into [HP/16][HP/16][16][16]
 * (k-chunk major), optionally transposed: the layout every GEMM of the kernels streams. */
int gaudi_host_pack_matrix(int H, int ldw, int col0, int HP, int transpose, const float* W, float* packed_out);
/* Split image of the same block (what the default 8-wave EDGE GEMMs stream; csrc/w8_split.h, round 5: fp16 pairs): packed_out
 * holds ceil(T/2) * T * 2 units of 1 KiB (T = HP/16), unit (m, t, p) = piece p of output tile t against the 32-input chunk m:
 * piece 0 = fp16(w * scale), piece 1 = fp16(w * scale - piece 0), round to nearest even (|w * scale - p0 - p1| <= 2^-22 |w * scale|
 * while piece 1 is a normal fp16 number; scale = a power of two, gaudi_host_weight_scale); lane L = (row L & 15, group g = L >> 4)
 * holds 8 fp16: slots 0-3 = inputs 16(2m) + 4g .. +3, slots 4-7 = inputs 16(2m+1) + 4g .. +3.  ktail != 0 and H % 16 == 4 with
 * an odd T >= 3: the last chunk holds the 4 tail inputs as T fp32 tiles of w * scale instead (float4 index (k % 16) * 16 + o % 16
 * of tile t, element 0). */
int gaudi_host_pack_matrix_split(int H, int ldw, int col0, int HP, int transpose, int ktail, float scale, const float* W,
                                 float* packed_out);
/* fp16-pair image of a NODE-GEMM matrix (csrc/w8_nodes_f16.h): packed_out holds HP * HP floats = (T / 2) * T * 2 units of 1 KiB,
 * unit (m, t, p) as above but lane L = (row L & 15, group g = L >> 4) holds inputs 32 m + 8 g .. +7, piece 0 = fp16(w * scale),
 * piece 1 = fp16((w * scale - piece 0) * 2^11); an odd T leaves the last 16 inputs as a trailing block of T * 256 fp32 values,
 * UNSCALED, [tile][k-step q][lane (row, g)] = w[row][16 (T - 1) + 4 q + g]. */
int gaudi_host_pack_matrix_f16(int H, int ldw, int col0, int HP, int transpose, float scale, const float* W, float* packed_out);
/* The power-of-two scale gaudi_load_edm / gaudi_load_predictor give a network's weight images: the largest finite |w| of the n
 * blocks (rows[i] x cols[i], row stride ldw[i]) lands in [2^13, 2^14).  0: refused -- an infinite weight, or a block whose largest
 * entry lies more than 2^17 below the largest of all (the blocks are judged as edge-level matrices; node-level ones may lie 2^25
 * below: round 6, through round 5 the rule was 2^12 for all) -- such a network runs the fp32-instruction kernels, and the load
 * leaves a warning (gaudi_last_warning). */
int gaudi_host_weight_scale(int n, const float* const* blocks, const int32_t* rows, const int32_t* cols, const int32_t* ldw,
                            float* scale_out);
/* LDS layout of the fp16-pair node GEMMs' activation copies (csrc/w8_nodes_f16.h; no reference counterpart: the reference's
 * node-level nn.Linear calls, egnn_new.py:42-89 / models.py:520-551, have no operand staging).  Float offset, inside the copy of
 * `column_tiles` tiles of 16 node columns, of the 16 bytes that hold inputs 32 chunk + 8 g .. + 7 of node column c (piece 0; piece 1
 * lies 256 floats further).  The layout is a performance contract the CPU suite checks: the 16 lanes (c = 0..15) of one g read 256
 * consecutive bytes, and the 32 lanes of one store pass of a row's split (4 chunks x 4 g x 2 halves) hit 64 different banks. */
int gaudi_host_node_operand_offset(int column_tiles, int chunk, int tile, int c, int g, int32_t* float_offset_out);

/* Kernel timing with HIP events on the handle's own stream (bench.py roofline).
 * gaudi_profile_reset enables collection; gaudi_profile_get returns the number of step-kernel
 * launches since the reset and their summed duration. */
int gaudi_profile_reset(gaudi_handle* h, int enable);
int gaudi_profile_get(gaudi_handle* h, int32_t* n_launches, double* total_ms, int64_t* steps_done);

/* Override the node count the predictor readout divides by (EGNN_predictor.forward takes the mean over
 * the PADDED N, egnn_predictor/models.py:457).  0 (default) = the N of each call. */
int gaudi_set_readout_nodes(gaudi_handle* h, int n_pad);

/* fix_noise=True of EnVariationalDiffusion.sample / sample_guidance (en_diffusion.py:562-566, 972-978, 1022-1028): while
 * enabled, every molecule of a gaudi_sample / gaudi_sample_cb / gaudi_sample_chain call receives the raw N(0,1) draws of ONE
 * sample -- the Philox stream of global sample index key_sample, or, with injected noise, a buffer [T+2][1][N][3+F] --
 * masked and mean-centred per molecule exactly as the reference broadcasts its [1,N,.] randn over the batch. */
int gaudi_set_fix_noise(gaudi_handle* h, int enable, int64_t key_sample);

/* Tuning knob: reverse steps fused into one kernel launch (default 25). */
int gaudi_set_steps_per_launch(gaudi_handle* h, int steps);

#ifdef __cplusplus
}
#endif
#endif /* GAUDI_HIP_H */
