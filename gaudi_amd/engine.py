"""Thin object wrapper over the C ABI: one Engine = one gaudi_handle = one GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TARGET_CB, TARGET_CBZ, Diag, EdmConfig, GaudiError, PredConfig, f32, fptr


def _strip(sd: dict) -> dict:
    """Checkpoints saved with dp=True carry a ``module.`` prefix (models_edm.py:98-102)."""
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[k[7:] if k.startswith("module.") else k] = f32(v)
    return out


def _noise_power(schedule: str) -> float:
    """-> gaudi_edm_config.noise_power: p of 'polynomial_<p>', 0 for 'cosine' (PredefinedNoiseSchedule, en_diffusion.py:191-202)."""
    if schedule == "cosine":
        return 0.0
    parts = str(schedule).split("_")
    try:
        power = float(parts[1]) if len(parts) == 2 and parts[0] == "polynomial" else 0.0
    except ValueError:  # 'polynomial_x'
        power = 0.0
    if not power > 0:
        raise GaudiError(f"unsupported diffusion_noise_schedule {schedule!r}: 'polynomial_<p>' and 'cosine' are implemented "
                         "(the 'learned' schedule is training-only in the reference)")
    return power


class Engine:
    def __init__(self, device: int = 0):
        self.lib = _lib.load_library()
        self.h = C.c_void_p()
        rc = self.lib.gaudi_create(int(device), C.byref(self.h))
        if rc != 0:
            raise GaudiError(f"gaudi_create(device={device}) failed with code {rc} (no usable HIP device?)")
        self.device = device
        self.edm_args = None
        self.pred_args = None
        self.fix_noise = False
        self.F = None
        self.K = None

    _default = None

    @classmethod
    def default(cls, device: int = 0) -> "Engine":
        """Process-wide handle for calls that need no weights (stability check); created on first use."""
        if cls._default is None or not cls._default.h:
            cls._default = cls(device)
        return cls._default

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.gaudi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _warning(self):
        """The library's last non-fatal condition ('' if none, or an older A/B library without the entry point)."""
        fn = getattr(self.lib, "gaudi_last_warning", None)
        if fn is None or fn.restype is not C.c_char_p:
            return ""
        msg = fn(self.h)
        return msg.decode() if msg else ""

    def _note_fallback(self):
        """A weight set the fp16-pair images refuse runs the fp32-instruction kernels at about 0.55 x the speed: say so once,
        loudly, and keep the reason for diag['edge_math_fallback']."""
        w = self._warning()
        if w and w != getattr(self, "_fallback_reason", None):
            import warnings
            warnings.warn("gaudi_amd: " + w, RuntimeWarning, stacklevel=3)
        if w:
            self._fallback_reason = w

    def profile_clock_mhz(self) -> float:
        """Shader clock the chip held during the most recent profiled sampler launch (0.0: none, or an older A/B library)."""
        fn = getattr(self.lib, "gaudi_profile_clock", None)
        if fn is None or not fn.argtypes:
            return 0.0
        v = C.c_double(0.0)
        self._check(fn(self.h, C.byref(v)), "gaudi_profile_clock")
        return float(v.value)

    def keep_h(self) -> int:
        """Floats of LDS the most recent call gave to a kept split copy of h (0: none)."""
        fn = getattr(self.lib, "gaudi_last_keep_h", None)
        if fn is None or not fn.argtypes:
            return 0
        n = C.c_int32(0)
        self._check(fn(self.h, C.byref(n)), "gaudi_last_keep_h")
        return int(n.value)

    def family_split(self) -> int:
        """Molecules of the most recent sample() call that ran on the resident kernels beside a V8G bucket (0: one family)."""
        fn = getattr(self.lib, "gaudi_last_family_split", None)
        if fn is None or not fn.argtypes:
            return 0
        n = C.c_int32(0)
        self._check(fn(self.h, C.byref(n)), "gaudi_last_family_split")
        return int(n.value)

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.gaudi_last_error(self.h)
            raise GaudiError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _tensor_args(self, sd: dict):
        names = list(sd.keys())
        arrs = [sd[k] for k in names]
        n = len(names)
        c_names = (C.c_char_p * n)(*[k.encode() for k in names])
        c_ptrs = (_lib.FP * n)(*[fptr(a) for a in arrs])
        c_numel = (C.c_int64 * n)(*[a.size for a in arrs])
        return n, c_names, c_ptrs, c_numel, arrs

    # ------------------------------------------------------------------ weights
    def load_edm(self, args: dict, state_dict: dict):
        """args: the checkpoint's args.txt namespace (utils/args_edm.py); state_dict: model.pt."""
        agg = args.get("aggregation_method", "sum")
        if agg not in ("sum", "mean"):
            raise GaudiError(f"unknown aggregation_method {agg!r}")
        sd = _strip(state_dict)
        emb = sd.get("dynamics.egnn.embedding.weight")
        if emb is None:
            raise GaudiError("state dict has no dynamics.egnn.embedding.weight (mode='gnn_dynamics' is unsupported)")
        F = emb.shape[1] - 1  # in_node_nf is not stored: recover it
        from .checkpoint import normalize_factors
        nv = normalize_factors(args)
        cfg = EdmConfig(F, int(args["nf"]), int(args["n_layers"]), int(args.get("inv_sublayers", 1)),
                        int(bool(args["attention"])), int(bool(args["tanh"])), float(args["coords_range"]),
                        float(args["norm_constant"]), float(args.get("normalization_factor", 1)) if agg == "sum" else 0.0,
                        int(args["diffusion_steps"]), _noise_power(args["diffusion_noise_schedule"]),
                        float(args["diffusion_noise_precision"]), (C.c_float * 3)(*[float(v) for v in nv]),
                        int(bool(args.get("sin_embedding", False))))
        n, c_names, c_ptrs, c_numel, keep = self._tensor_args(sd)
        self._check(self.lib.gaudi_load_edm(self.h, C.byref(cfg), n, c_names, c_ptrs, c_numel), "gaudi_load_edm")
        self._note_fallback()
        self.edm_args, self.F, self.T = dict(args), F, int(args["diffusion_steps"])
        # EnVariationalDiffusion.check_issues_norm_values (en_diffusion.py:336-349): the reference refuses to BUILD a model whose
        # sigma_0 is not small against 1 / norm_value (8 standard deviations) -- e.g. 'cosine' with the default
        # normalize_factors [3, 4, 10]; a checkpoint it could never have produced is refused here as well
        sigma_0 = float(np.sqrt(1.0 / (1.0 + np.exp(-np.float64(self.gamma()[0])))))
        max_norm = max(float(nv[1]), float(nv[2]))
        if sigma_0 * 8 > 1.0 / max_norm:
            raise GaudiError(f"Value for normalization value {max_norm} probably too large with sigma_0 {sigma_0:.5f} and "
                             f"1 / norm_value = {1.0 / max_norm} (en_diffusion.py:336-349: the reference raises the same)")

    def load_predictor(self, args: dict, state_dict: dict):
        sd = _strip(state_dict)
        emb = sd["egnn.embedding.weight"]
        F = emb.shape[1] - 1
        K = sd["egnn.embedding_out.weight"].shape[0]
        cfg = PredConfig(F, K, int(args["nf"]), int(args["n_layers"]), int(bool(args["attention"])),
                         int(bool(args["tanh"])), float(args["coords_range"]))
        n, c_names, c_ptrs, c_numel, keep = self._tensor_args(sd)
        self._check(self.lib.gaudi_load_predictor(self.h, C.byref(cfg), n, c_names, c_ptrs, c_numel),
                    "gaudi_load_predictor")
        self._note_fallback()
        self.pred_args, self.K = dict(args), K

    # ------------------------------------------------------------------ tables
    def gamma(self) -> np.ndarray:
        g = np.empty(self.T + 1, np.float32)
        self._check(self.lib.gaudi_get_gamma(self.h, fptr(g)), "gaudi_get_gamma")
        return g

    def step_coefficients(self) -> np.ndarray:
        c = np.empty((self.T, 4), np.float32)
        self._check(self.lib.gaudi_get_step_coefficients(self.h, fptr(c)), "gaudi_get_step_coefficients")
        return c

    # ------------------------------------------------------------------ unit entry points
    @staticmethod
    def _masks(node_mask, edge_mask, B, N):
        nm = f32(node_mask).reshape(B, N)
        em = f32(edge_mask).reshape(B, N, N)
        return nm, em

    def phi(self, z, t, node_mask, edge_mask) -> np.ndarray:
        z = f32(z)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        t = f32(np.broadcast_to(np.asarray(t, np.float32).reshape(-1), (B,)))
        out = np.empty_like(z)
        self._check(self.lib.gaudi_phi(self.h, B, N, fptr(z), fptr(t), fptr(nm), fptr(em), fptr(out)), "gaudi_phi")
        return out

    def predictor_fwd(self, z, t, node_mask, edge_mask) -> np.ndarray:
        z = f32(z)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        t = f32(np.broadcast_to(np.asarray(t, np.float32).reshape(-1), (B,)))
        out = np.empty((B, self.K), np.float32)
        self._check(self.lib.gaudi_predictor_fwd(self.h, B, N, fptr(z), fptr(t), fptr(nm), fptr(em), fptr(out)),
                    "gaudi_predictor_fwd")
        return out

    def predictor_grad(self, z, t, node_mask, edge_mask, dpred):
        z = f32(z)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        t = f32(np.broadcast_to(np.asarray(t, np.float32).reshape(-1), (B,)))
        dp = f32(np.broadcast_to(np.asarray(dpred, np.float32), (B, self.K)))
        pred = np.empty((B, self.K), np.float32)
        grad = np.empty_like(z)
        self._check(self.lib.gaudi_predictor_grad(self.h, B, N, fptr(z), fptr(t), fptr(nm), fptr(em), fptr(dp),
                                                  fptr(pred), fptr(grad)), "gaudi_predictor_grad")
        return pred, grad

    def step(self, s_idx, z_t, node_mask, edge_mask, eps_raw, target_w=None, scale=1.0) -> np.ndarray:
        z = f32(z_t)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        eps = f32(eps_raw)
        tw = None if target_w is None else f32(target_w)
        out = np.empty_like(z)
        self._check(self.lib.gaudi_step(self.h, B, N, int(s_idx), fptr(z), fptr(nm), fptr(em), fptr(eps), fptr(tw),
                                        float(scale), fptr(out)), "gaudi_step")
        return out

    def decode(self, z0, node_mask, edge_mask, eps_raw):
        z = f32(z0)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        eps = f32(eps_raw)
        x = np.empty((B, N, 3), np.float32)
        h = np.empty((B, N, D - 3), np.float32)
        self._check(self.lib.gaudi_decode(self.h, B, N, fptr(z), fptr(nm), fptr(em), fptr(eps), fptr(x), fptr(h)),
                    "gaudi_decode")
        return x, h

    # ------------------------------------------------------------------ whole chain
    def sample(self, node_mask, edge_mask, *, seed=0, sample_offset=0, noise=None, std=1.0, target_w=None, scale=1.0,
               return_z0=False):
        nm = f32(node_mask)
        B, N = nm.shape[0], nm.shape[1]
        nm, em = self._masks(nm, edge_mask, B, N)
        D = 3 + self.F
        nz = None
        if noise is not None:
            nz = f32(noise)
            want = (self.T + 2, 1 if self.fix_noise else B, N, D)
            if nz.shape != want:
                raise GaudiError(f"noise must be [T+2,{'1' if self.fix_noise else 'B'},N,3+F] = {want}, got {nz.shape}")
        tw = None if target_w is None else f32(target_w)
        x = np.empty((B, N, 3), np.float32)
        h = np.empty((B, N, self.F), np.float32)
        z0 = np.empty((B, N, D), np.float32) if return_z0 else None
        diag = Diag()
        self._check(self.lib.gaudi_sample(self.h, B, N, fptr(nm), fptr(em), int(seed), int(sample_offset), fptr(nz),
                                          float(std), fptr(tw), float(scale), fptr(x), fptr(h), fptr(z0),
                                          C.byref(diag)), "gaudi_sample")
        d = dict(max_masked_leak=diag.max_masked_leak, max_cog_rel=diag.max_cog_rel, max_cog_abs=diag.max_cog_abs,
                 nan_count=diag.nan_count, reprojected=diag.reprojected,
                 edge_math_fallback=getattr(self, "_fallback_reason", None), family_split_resident=self.family_split())
        return (x, h, d, z0) if return_z0 else (x, h, d)

    def predict_noised(self, x, onehot, t_int, node_mask, edge_mask, *, seed=0, sample_offset=0, noise=None):
        """sample_edm_t + predictor forward in one launch -> (z_t [B,N,3+F], pred [B,K])."""
        x = f32(x)
        B, N = x.shape[0], x.shape[1]
        oh = f32(onehot).reshape(B, N, self.F)
        nm, em = self._masks(f32(node_mask).reshape(B, N), edge_mask, B, N)
        ti = np.ascontiguousarray(np.broadcast_to(np.asarray(t_int).reshape(-1), (B,)), dtype=np.int32)
        nz = None
        if noise is not None:
            nz = f32(noise)
            if nz.shape != (B, N, 3 + self.F):
                raise GaudiError(f"noise must be [B,N,3+F] = {(B, N, 3 + self.F)}, got {nz.shape}")
        zt = np.empty((B, N, 3 + self.F), np.float32)
        pred = np.empty((B, self.K), np.float32)
        self._check(self.lib.gaudi_predict_noised(self.h, B, N, fptr(x), fptr(oh), ti.ctypes.data_as(_lib.IP), fptr(nm),
                                                  fptr(em), int(seed), int(sample_offset), fptr(nz), fptr(zt),
                                                  fptr(pred)), "gaudi_predict_noised")
        return zt, pred

    def sample_callback(self, node_mask, edge_mask, target_grad, *, seed=0, sample_offset=0, noise=None, std=1.0,
                        scale=1.0, return_z0=False, with_z=False):
        """Guided chain for an arbitrary target: ``target_grad(pred [B,K], t) -> dT/dpred [B,K]`` is called once per
        reverse step between the two device phases (include/gaudi_hip.h: gaudi_sample_cb).  with_z=True: the target also
        depends on z outside the predictor -- ``target_grad(z_s [B,N,D], pred [B,K], t) -> (dT/dpred [B,K], dT/dz [B,N,D])``
        with dT/dz the DIRECT part, pred held fixed (gaudi_sample_cbz)."""
        nm = f32(node_mask)
        B, N = nm.shape[0], nm.shape[1]
        nm, em = self._masks(nm, edge_mask, B, N)
        D = 3 + self.F
        nz = None
        if noise is not None:
            nz = f32(noise)
            want = (self.T + 2, 1 if self.fix_noise else B, N, D)
            if nz.shape != want:
                raise GaudiError(f"noise must be [T+2,{'1' if self.fix_noise else 'B'},N,3+F] = {want}, got {nz.shape}")
        x = np.empty((B, N, 3), np.float32)
        h = np.empty((B, N, self.F), np.float32)
        z0 = np.empty((B, N, D), np.float32) if return_z0 else None
        diag = Diag()
        failure = []

        def _cb(_user, b, k, pred_p, t, out_p):
            try:
                pred = np.ctypeslib.as_array(pred_p, shape=(b, k)).copy()
                g = np.ascontiguousarray(target_grad(pred, float(t)), dtype=np.float32)
                if g.shape != (b, k):
                    raise GaudiError(f"target_grad must return [B,K] = {(b, k)}, got {g.shape}")
                np.ctypeslib.as_array(out_p, shape=(b, k))[...] = g
            except BaseException as exc:  # never unwind through the C frames; report after the call returns
                if not failure:
                    failure.append(exc)

        def _cbz(_user, b, n, dd, k, z_p, pred_p, t, out_p, outz_p):
            try:
                zs = np.ctypeslib.as_array(z_p, shape=(b, n, dd)).copy()
                pred = np.ctypeslib.as_array(pred_p, shape=(b, k)).copy()
                g, gz = target_grad(zs, pred, float(t))
                g = np.ascontiguousarray(g, dtype=np.float32)
                gz = np.ascontiguousarray(gz, dtype=np.float32)
                if g.shape != (b, k) or gz.shape != (b, n, dd):
                    raise GaudiError(f"target_grad must return ([B,K], [B,N,D]) = ({(b, k)}, {(b, n, dd)}), got ({g.shape}, {gz.shape})")
                np.ctypeslib.as_array(out_p, shape=(b, k))[...] = g
                np.ctypeslib.as_array(outz_p, shape=(b, n, dd))[...] = gz
            except BaseException as exc:  # never unwind through the C frames; report after the call returns
                if not failure:
                    failure.append(exc)

        if with_z:
            cb = TARGET_CBZ(_cbz)
            rc = self.lib.gaudi_sample_cbz(self.h, B, N, fptr(nm), fptr(em), int(seed), int(sample_offset), fptr(nz),
                                           float(std), cb, None, float(scale), fptr(x), fptr(h), fptr(z0), C.byref(diag))
        else:
            cb = TARGET_CB(_cb)
            rc = self.lib.gaudi_sample_cb(self.h, B, N, fptr(nm), fptr(em), int(seed), int(sample_offset), fptr(nz),
                                          float(std), cb, None, float(scale), fptr(x), fptr(h), fptr(z0), C.byref(diag))
        if failure:
            raise failure[0]
        self._check(rc, "gaudi_sample_cbz" if with_z else "gaudi_sample_cb")
        d = dict(max_masked_leak=diag.max_masked_leak, max_cog_rel=diag.max_cog_rel, max_cog_abs=diag.max_cog_abs,
                 nan_count=diag.nan_count, reprojected=diag.reprojected,
                 edge_math_fallback=getattr(self, "_fallback_reason", None), family_split_resident=self.family_split())
        return (x, h, d, z0) if return_z0 else (x, h, d)

    def sample_chain(self, node_mask, edge_mask, keep_frames, *, seed=0, sample_offset=0, noise=None, std=1.0):
        """-> chain [keep_frames, B, N, 3+F] (frame 0 = final [x | one_hot])."""
        nm = f32(node_mask)
        B, N = nm.shape[0], nm.shape[1]
        nm, em = self._masks(nm, edge_mask, B, N)
        D = 3 + self.F
        nz = None if noise is None else f32(noise)
        chain = np.empty((int(keep_frames), B, N, D), np.float32)
        self._check(self.lib.gaudi_sample_chain(self.h, B, N, fptr(nm), fptr(em), int(seed), int(sample_offset), fptr(nz),
                                                float(std), int(keep_frames), fptr(chain)), "gaudi_sample_chain")
        return chain

    def philox_normal(self, seed, sample_offset, B, n_elem, draw0, n_draws) -> np.ndarray:
        out = np.empty((n_draws, B, n_elem), np.float32)
        self._check(self.lib.gaudi_philox_normal(self.h, int(seed), int(sample_offset), B, n_elem, draw0, n_draws,
                                                 fptr(out)), "gaudi_philox_normal")
        return out

    # ------------------------------------------------------------------ profiling
    def profile_reset(self, enable=True):
        self._check(self.lib.gaudi_profile_reset(self.h, int(enable)), "gaudi_profile_reset")

    def profile_get(self):
        n = C.c_int32()
        ms = C.c_double()
        steps = C.c_int64()
        self._check(self.lib.gaudi_profile_get(self.h, C.byref(n), C.byref(ms), C.byref(steps)), "gaudi_profile_get")
        return n.value, ms.value, steps.value

    def stability_profile_get(self):
        n = C.c_int32()
        ms = C.c_double()
        self._check(self.lib.gaudi_stability_profile_get(self.h, C.byref(n), C.byref(ms)), "gaudi_stability_profile_get")
        return n.value, ms.value

    def set_fix_noise(self, enable: bool, key_sample: int = 0):
        """fix_noise=True of the reference (en_diffusion.py:562-566): every molecule of a call receives the raw draws of
        ONE sample (Philox stream of global sample ``key_sample``, or injected noise of shape [T+2,1,N,3+F])."""
        self._check(self.lib.gaudi_set_fix_noise(self.h, int(bool(enable)), int(key_sample)), "gaudi_set_fix_noise")
        self.fix_noise = bool(enable)

    def kernel_variant(self):
        """-> (configured, last_call): 8 = eight waves per molecule (default), 4 = four waves (GAUDI_WAVES=4 or fallback)."""
        a, b = C.c_int32(), C.c_int32()
        self._check(self.lib.gaudi_kernel_variant(self.h, C.byref(a), C.byref(b)), "gaudi_kernel_variant")
        return a.value, b.value

    def edge_math(self):
        """-> (configured, last_call): 1 = edge GEMMs on the bf16 matrix pipe with exactly split fp32 operands (default on the
        8-wave kernels; last_call 2 = the same with the half-size LDS weight ring), 0 = fp32 matrix instructions
        (GAUDI_EDGE_MATH=fp32, the 4-wave kernels, or the LDS fallback)."""
        a, b = C.c_int32(), C.c_int32()
        self._check(self.lib.gaudi_edge_math(self.h, C.byref(a), C.byref(b)), "gaudi_edge_math")
        return a.value, b.value

    def node_buffers_global(self) -> bool:
        """True if the most recent call kept its node buffers in a global scratch (molecules beyond the LDS limit: the V8G kernels
        on 8 waves, the V4G kernels on 4 -- kernel_variant() tells which)."""
        v = C.c_int32()
        self._check(self.lib.gaudi_node_buffers(self.h, C.byref(v)), "gaudi_node_buffers")
        return v.value in (1, 2)

    def node_buffers_form(self) -> int:
        """0: the most recent call kept its node buffers in LDS; 1: in a global scratch; 2: in a global scratch except P and Q, which
        stayed in LDS (8-wave kernels, round 6: kern8gp_*.hip -- taken where that plan fits); 3: in LDS except ONE of the predictor's
        five (wide groups on the full weight ring, round 6: kern8mp_fused.hip)."""
        v = C.c_int32()
        self._check(self.lib.gaudi_node_buffers(self.h, C.byref(v)), "gaudi_node_buffers")
        return int(v.value)

    def last_workgroups(self) -> int:
        """Workgroups of the most recent launch (molecules, or the groups they were packed into)."""
        return self.last_launch_shape()[0]

    def last_launch_shape(self):
        """-> (workgroups, node slots per workgroup) of the most recent launch; node slots > N: wide groups."""
        n, ns = C.c_int32(), C.c_int32()
        self._check(self.lib.gaudi_last_workgroups(self.h, C.byref(n), C.byref(ns)), "gaudi_last_workgroups")
        return n.value, ns.value

    def set_plan_hint(self, min_slots: int = 0, force_waves: int = 0):
        """Plan the following calls with the graph figures of a larger logical batch (see gaudi_set_plan_hint)."""
        self._check(self.lib.gaudi_set_plan_hint(self.h, int(min_slots), int(force_waves)), "gaudi_set_plan_hint")
        self._plan_hint = (int(min_slots), int(force_waves))

    def plan_hint_for(self, node_mask, edge_mask):
        """(min_slots, force_waves) of a WHOLE logical batch: what every shard of it passes to set_plan_hint so that all
        shards run the same kernel family and edge-GEMM arithmetic (device-free: gaudi_host_graph_meta8)."""
        nm = f32(node_mask)
        B, N = nm.shape[0], nm.shape[1]
        nm, em = self._masks(nm, edge_mask, B, N)
        slots = C.c_int32()
        rc = self.lib.gaudi_host_graph_meta8(B, N, fptr(nm), fptr(em), C.byref(slots), None, None, None, None, None, None,
                                             None, 0, None)
        if rc == -5:  # GAUDI_E_CAPACITY: a node with more than 32 live edges -> the 4-wave kernels for everybody
            return 0, 4
        if rc != 0:
            raise GaudiError(f"gaudi_host_graph_meta8 failed ({rc})")
        return int(slots.value), 0

    def pack_plan(self, node_mask, edge_mask, node_slots=None, tiles=16):
        """-> (G, group_of[B], ntiles[G], ncols[G]): the workgroups a sampling call of the 8-wave kernels launches for this
        batch -- small molecules share a workgroup as components of one disjoint graph (device-free: gaudi_host_pack_plan;
        G = B when nothing packs).  node_slots > N: the plan of WIDE groups (gaudi_host_pack_plan_wide)."""
        nm = f32(node_mask)
        B, N = nm.shape[0], nm.shape[1]
        nm, em = self._masks(nm, edge_mask, B, N)
        G = C.c_int32()
        group_of, ntiles, ncols = (np.zeros(B, np.int32) for _ in range(3))
        i32 = C.POINTER(C.c_int32)
        if node_slots is not None and node_slots > N:
            rc = self.lib.gaudi_host_pack_plan_wide(B, N, int(node_slots), int(tiles), fptr(nm), fptr(em), C.byref(G),
                                                    group_of.ctypes.data_as(i32), ntiles.ctypes.data_as(i32), ncols.ctypes.data_as(i32))
        else:
            rc = self.lib.gaudi_host_pack_plan(B, N, fptr(nm), fptr(em), C.byref(G), group_of.ctypes.data_as(i32),
                                               ntiles.ctypes.data_as(i32), ncols.ctypes.data_as(i32))
        if rc != 0:
            raise GaudiError(f"gaudi_host_pack_plan failed ({rc})")
        return G.value, group_of, ntiles[:G.value], ncols[:G.value]

    def set_steps_per_launch(self, k: int):
        self._check(self.lib.gaudi_set_steps_per_launch(self.h, int(k)), "gaudi_set_steps_per_launch")
