"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- numpy restatement of the GaUDI
reverse-diffusion sampler: EDM denoiser, conditional predictor (+ analytic input
gradient), unguided / guided reverse step, final decode and the two sampling loops.

Every function cites the reference file:line (relative to the upstream repo root) it
follows.  The arithmetic is written "as the reference writes it" (concat -> Linear, dense
N x N edge set incl. self loops, multiply by masks) so that differences against the
reference are pure BLAS summation-order noise (~1e-6 rel).  ``dtype`` may be switched to
float64 to measure the conditioning of long chains.

State dicts use the reference's parameter names without the ``module.`` prefix
("checkpoint / resume").
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _sigmoid(x):
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-x))


def _silu(x):
    return x * _sigmoid(x)


def _dsilu(x):
    s = _sigmoid(x)
    return s * (1.0 + x * (1.0 - s))


def _linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def _softplus(x):
    # torch.nn.functional.softplus (beta=1, threshold=20)
    x = np.asarray(x)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def _logsigmoid(x):
    # torch: min(x,0) - log1p(exp(-|x|))
    x = np.asarray(x)
    return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def strip_module_prefix(sd):
    """models_edm.py:98-102: checkpoints saved with dp=True carry a ``module.`` prefix."""
    out = {}
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = np.asarray(v)
    return out


# --------------------------------------------------------------------------------------
# noise schedule  (edm/equivariant_diffusion/en_diffusion.py:32-61, 186-230)
# --------------------------------------------------------------------------------------
def clip_noise_schedule(alphas2, clip_value=0.001):
    """en_diffusion.py:32-44."""
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    alphas_step = alphas2[1:] / alphas2[:-1]
    alphas_step = np.clip(alphas_step, a_min=clip_value, a_max=1.0)
    return np.cumprod(alphas_step, axis=0)


def polynomial_schedule(timesteps, s=1e-4, power=3.0):
    """en_diffusion.py:47-61 (float64 numpy, as the reference)."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(x / steps, power)) ** 2
    alphas2 = clip_noise_schedule(alphas2, clip_value=0.001)
    precision = 1 - 2 * s
    return precision * alphas2 + s


def cosine_beta_schedule(timesteps, s=0.008):
    """en_diffusion.py:64-81 (raise_to_power = 1) -> alphas_cumprod [T+1]."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    return np.cumprod(1.0 - betas, axis=0)


def gamma_table(noise_schedule, timesteps, precision):
    """PredefinedNoiseSchedule.__init__, en_diffusion.py:191-218 -> float32 [T+1]."""
    if noise_schedule == "cosine":
        alphas2 = cosine_beta_schedule(timesteps)
    elif "polynomial" in noise_schedule:
        splits = noise_schedule.split("_")
        assert len(splits) == 2
        alphas2 = polynomial_schedule(timesteps, s=precision, power=float(splits[1]))
    else:
        raise ValueError(f"unsupported noise schedule {noise_schedule!r}")
    sigmas2 = 1 - alphas2
    return (-(np.log(alphas2) - np.log(sigmas2))).astype(F32)


def step_coefficients(gamma, s_idx, t_idx, dtype=F32):
    """Scalars of one reverse step, en_diffusion.py:365-373, 433-457, 811-821, 843-849.

    Returns dict(inv_alpha_ts, eps_coef, sigma, sigma_s, sigma_t, alpha_ts, sigma2_ts).
    mu = z_t * inv... is NOT how the reference writes it: it divides, so we return alpha_ts
    and let the caller divide.
    """
    g_s = gamma[s_idx].astype(dtype)
    g_t = gamma[t_idx].astype(dtype)
    sigma2_ts = -np.expm1(_softplus(g_s) - _softplus(g_t))
    log_a2_ts = _logsigmoid(-g_t) - _logsigmoid(-g_s)
    alpha_ts = np.exp(dtype(0.5) * log_a2_ts)
    sigma_ts = np.sqrt(sigma2_ts)
    sigma_s = np.sqrt(_sigmoid(g_s))
    sigma_t = np.sqrt(_sigmoid(g_t))
    return dict(
        alpha_ts=dtype(alpha_ts),
        sigma2_ts=dtype(sigma2_ts),
        eps_coef=dtype(dtype(sigma2_ts) / dtype(alpha_ts) / dtype(sigma_t)),
        sigma=dtype(dtype(sigma_ts) * dtype(sigma_s) / dtype(sigma_t)),
        sigma_s=dtype(sigma_s),
        sigma_t=dtype(sigma_t),
    )


# --------------------------------------------------------------------------------------
# masks  (sampling_edm.py:119-161, 172-209)
# --------------------------------------------------------------------------------------
def node2edge_mask(node_mask):
    """sampling_edm.py:119-125.  node_mask [B,N] -> [B,N,N] with zero diagonal."""
    em = node_mask[:, None, :] * node_mask[:, :, None]
    em = em * (1.0 - np.eye(node_mask.shape[1], dtype=node_mask.dtype))[None]
    return em


def build_masks(nodesxsample, max_nodes, orientation):
    """sampling_edm.py:135-161 / 176-209.

    Returns node_mask [B,N,1], edge_mask [B*N*N,1] (N doubled when ``orientation``).  The
    identity blocks that tie ring i to orientation node i are NOT masked for padded rings
    (reference quirk, sampling_edm.py:147-159).
    """
    nodesxsample = np.asarray(nodesxsample).astype(np.int64)
    B = len(nodesxsample)
    nm = np.zeros((B, max_nodes), dtype=F32)
    for i in range(B):
        nm[i, : nodesxsample[i]] = 1
    em = node2edge_mask(nm)
    n = max_nodes
    if orientation:
        eye = np.broadcast_to(np.eye(n, dtype=F32), (B, n, n))
        top = np.concatenate([em, eye], axis=1)  # [B,2n,n]
        right = np.broadcast_to(
            np.concatenate([np.eye(n, dtype=F32), np.zeros((n, n), dtype=F32)], axis=0), (B, 2 * n, n)
        )
        em = np.concatenate([top, right], axis=2)  # [B,2n,2n]
        nm = np.concatenate([nm, nm], axis=1)
        n *= 2
    return nm[:, :, None].copy(), np.ascontiguousarray(em).reshape(-1, 1)


def remove_mean_with_mask(x, node_mask):
    """edm/equivariant_diffusion/utils.py:33-44."""
    N = np.maximum(node_mask.sum(1, keepdims=True), 1)
    mean = x.sum(1, keepdims=True) / N
    return x - mean * node_mask


# --------------------------------------------------------------------------------------
# EDM denoiser  (edm/egnn/models.py:83-152, edm/egnn/egnn_new.py)
# --------------------------------------------------------------------------------------
def _coord2diff(x, norm_constant):
    """egnn_new.py:394-400 on the dense edge set: x [B,N,3] -> radial [B,N,N,1], diff [B,N,N,3]."""
    diff = x[:, :, None, :] - x[:, None, :, :]
    radial = (diff**2).sum(-1, keepdims=True)
    norm = np.sqrt(radial + x.dtype.type(1e-8))
    return radial, diff / (norm + x.dtype.type(norm_constant))


def sin_frequencies(dtype=F32):
    """SinusoidsEmbeddingNew.__init__ (egnn_new.py:378-385): 2 pi 4^k / 15, k = 0..5.  torch builds them as
    ``2 * math.pi * div_factor ** torch.arange(n) / max_res``: the Python double times an int64 tensor gives a float32 tensor (the
    product rounded to fp32 -- exact here: a power of four), then the fp32 division by 15."""
    n = int(np.log(15.0 / (15.0 / 2000.0)) / np.log(4.0)) + 1  # 6
    k = 4.0 ** np.arange(n)
    # (the tensor is neither a parameter nor a buffer: model.double() leaves it in fp32, and the float64 run of the reference
    # multiplies by the fp32-rounded frequencies -- the float64 evaluation here does the same)
    return ((np.float32(2.0 * np.pi) * k.astype(np.float32)) / np.float32(15.0)).astype(dtype)


def _sin_embedding(x, dtype):
    """SinusoidsEmbeddingNew.forward (egnn_new.py:387-391) on [..., 1] -> [..., 12]: sqrt(x + 1e-8) * f, (sin | cos)."""
    e = np.sqrt(x + dtype(1e-8)) * sin_frequencies(dtype)
    return np.concatenate([np.sin(e), np.cos(e)], axis=-1).astype(dtype)


def _edge_input(h, edge_attr):
    B, N, H = h.shape
    hi = np.broadcast_to(h[:, :, None, :], (B, N, N, H))
    hj = np.broadcast_to(h[:, None, :, :], (B, N, N, H))
    return np.concatenate([hi, hj, edge_attr], axis=-1)


def edm_phi(sd, cfg, z, t, node_mask, edge_mask, dtype=F32):
    """EGNN_dynamics._forward (edm/egnn/models.py:83-152) -> eps_hat [B,N,3+F].

    z [B,N,3+F]; t scalar or [B]; node_mask [B,N,1]; edge_mask anything reshapeable to
    [B,N,N,1].  cfg keys: n_layers, inv_sublayers, attention, tanh, coords_range,
    norm_constant, normalization_factor.
    """
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if k.startswith("dynamics.")}
    z = np.asarray(z, dtype=dtype)
    B, N, D = z.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    em = np.asarray(edge_mask, dtype=dtype).reshape(B, N, N, 1)
    xh = z * nm  # models.py:90
    x = xh[:, :, :3].copy()
    h = xh[:, :, 3:]
    tt = np.broadcast_to(np.asarray(t, dtype=dtype).reshape(-1, 1, 1), (B, N, 1))
    h = np.concatenate([h, tt], axis=2)  # time column is NOT masked (models.py:97-105)

    p = "dynamics.egnn."
    d0, _ = _coord2diff(x, 1.0)  # egnn_new.py:301 (radial only)
    sin_emb = bool(cfg.get("sin_embedding", False))
    if sin_emb:
        d0 = _sin_embedding(d0, dtype)  # egnn_new.py:302-303
    h = _linear(h, sd[p + "embedding.weight"], sd[p + "embedding.bias"])
    x_in = x
    # unsorted_segment_sum (egnn_new.py:403-421): 'sum' divides by normalization_factor; 'mean' by the number of edges of the
    # DENSE list that share the row -- masked ones included (their contribution is zero, their count is not) -- i.e. by the
    # padded node count N
    agg_method = cfg.get("aggregation_method", "sum")
    assert agg_method in ("sum", "mean")
    normf = dtype(cfg["normalization_factor"]) if agg_method == "sum" else dtype(N)
    for l in range(cfg["n_layers"]):
        bp = f"{p}e_block_{l}."
        radial, cdiff = _coord2diff(x, cfg["norm_constant"])  # egnn_new.py:216
        if sin_emb:
            radial = _sin_embedding(radial, dtype)  # :217-218
        edge_attr = np.concatenate([radial, d0], axis=-1)  # :219
        for s in range(cfg["inv_sublayers"]):
            gp = f"{bp}gcl_{s}."
            inp = _edge_input(h, edge_attr)
            mij = _silu(_linear(inp, sd[gp + "edge_mlp.0.weight"], sd[gp + "edge_mlp.0.bias"]))
            mij = _silu(_linear(mij, sd[gp + "edge_mlp.2.weight"], sd[gp + "edge_mlp.2.bias"]))
            if cfg["attention"]:
                att = _sigmoid(_linear(mij, sd[gp + "att_mlp.0.weight"], sd[gp + "att_mlp.0.bias"]))
                ef = mij * att
            else:
                ef = mij
            ef = ef * em
            agg = ef.sum(axis=2) / normf  # egnn_new.py:403-414 (row = i)
            nin = np.concatenate([h, agg], axis=-1)
            out = _silu(_linear(nin, sd[gp + "node_mlp.0.weight"], sd[gp + "node_mlp.0.bias"]))
            out = _linear(out, sd[gp + "node_mlp.2.weight"], sd[gp + "node_mlp.2.bias"])
            h = (h + out) * nm
        ep = f"{bp}gcl_equiv."
        inp = _edge_input(h, edge_attr)
        c = _silu(_linear(inp, sd[ep + "coord_mlp.0.weight"], sd[ep + "coord_mlp.0.bias"]))
        c = _silu(_linear(c, sd[ep + "coord_mlp.2.weight"], sd[ep + "coord_mlp.2.bias"]))
        phi = _linear(c, sd[ep + "coord_mlp.4.weight"])
        if cfg["tanh"]:
            # coords_range is the raw argument, not divided by n_layers (egnn_new.py:290)
            trans = cdiff * np.tanh(phi) * dtype(cfg["coords_range"])
        else:
            trans = cdiff * phi
        trans = trans * em
        x = (x + trans.sum(axis=2) / normf) * nm
        h = h * nm
    h = _linear(h, sd[p + "embedding_out.weight"], sd[p + "embedding_out.bias"]) * nm
    vel = (x - x_in) * nm
    h_final = h[:, :, :-1]  # drop time column (models.py:132-134)
    vel = np.nan_to_num(vel, nan=0.0) if np.isnan(vel).any() else vel
    vel = remove_mean_with_mask(vel, nm)
    return np.concatenate([vel, h_final], axis=2)


# --------------------------------------------------------------------------------------
# conditional predictor  (edm/egnn_predictor/models.py:433-457, 543-560; gcl.py:225-316)
# --------------------------------------------------------------------------------------
def _pred_cfg_range(cfg):
    return float(cfg["coords_range"]) / cfg["n_layers"]  # egnn_predictor/models.py:515


def predictor_forward(sd, cfg, z, node_mask, edge_mask, t, dtype=F32, return_cache=False):
    """EGNN_predictor.forward -> pred [B,K].  Readout divides by the PADDED N (models.py:457)."""
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    z = np.asarray(z, dtype=dtype)
    B, N, D = z.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    em = np.asarray(edge_mask, dtype=dtype).reshape(B, N, N, 1)
    x = z[:, :, :3] * nm
    h = z[:, :, 3:] * nm
    tt = np.broadcast_to(np.asarray(t, dtype=dtype).reshape(-1, 1, 1), (B, N, 1))
    h0 = np.concatenate([h, tt], axis=2)
    diff0 = x[:, :, None, :] - x[:, None, :, :]
    d0 = (diff0**2).sum(-1, keepdims=True)  # models.py:452
    p = "egnn."
    h = _linear(h0, sd[p + "embedding.weight"], sd[p + "embedding.bias"])
    R = dtype(_pred_cfg_range(cfg))
    cache = dict(x0=x, h0=h0, d0=d0, layers=[])
    for l in range(cfg["n_layers"]):
        gp = f"{p}gcl_{l}."
        radial, cdiff = _coord2diff(x, 1.0)  # gcl.py:308-316
        inp = _edge_input(h, np.concatenate([radial, d0], axis=-1))
        u = _linear(inp, sd[gp + "edge_mlp.0.weight"], sd[gp + "edge_mlp.0.bias"])
        v = _linear(_silu(u), sd[gp + "edge_mlp.2.weight"], sd[gp + "edge_mlp.2.bias"])
        m = _silu(v)
        if cfg["attention"]:
            a = _sigmoid(_linear(m, sd[gp + "att_mlp.0.weight"], sd[gp + "att_mlp.0.bias"]))
        else:
            a = np.ones_like(m[..., :1])
        e = m * a * em
        cpre = _linear(e, sd[gp + "coord_mlp.0.weight"], sd[gp + "coord_mlp.0.bias"])
        phi = _linear(_silu(cpre), sd[gp + "coord_mlp.2.weight"])
        tau = np.tanh(phi) * R if cfg["tanh"] else phi
        trans = cdiff * tau * em
        x_new = (x + trans.sum(axis=2)) * nm
        agg = e.sum(axis=2)
        npre = _linear(np.concatenate([h, agg], axis=-1), sd[gp + "node_mlp.0.weight"], sd[gp + "node_mlp.0.bias"])
        h_new = (h + _linear(_silu(npre), sd[gp + "node_mlp.2.weight"], sd[gp + "node_mlp.2.bias"])) * nm
        if return_cache:
            cache["layers"].append(dict(h=h, x=x, u=u, v=v, m=m, a=a, e=e, cpre=cpre, phi=phi, tau=tau,
                                        cdiff=cdiff, radial=radial, npre=npre))
        h, x = h_new, x_new
    hout = _linear(h, sd[p + "embedding_out.weight"], sd[p + "embedding_out.bias"]) * nm
    pred = hout.mean(axis=1)  # over padded N
    if return_cache:
        return pred, cache
    return pred


def predictor_grad(sd, cfg, z, node_mask, edge_mask, t, dpred, dtype=F32):
    """d(sum_b dpred[b] . pred[b]) / dz  -- hand-written reverse pass replacing
    ``torch.autograd.grad`` at en_diffusion.py:900-903 (weights frozen, utils/helpers.py:198-202).

    dpred [B,K] is dT/dpred (already multiplied by ``scale``).  Returns (pred, grad[B,N,3+F]).
    """
    pred, C = predictor_forward(sd, cfg, z, node_mask, edge_mask, t, dtype=dtype, return_cache=True)
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    z = np.asarray(z, dtype=dtype)
    B, N, D = z.shape
    H = sd["egnn.embedding.weight"].shape[0]
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    em = np.asarray(edge_mask, dtype=dtype).reshape(B, N, N, 1)
    dpred = np.asarray(dpred, dtype=dtype).reshape(B, -1)
    p = "egnn."
    R = dtype(_pred_cfg_range(cfg))
    one = dtype(1.0)
    # readout: pred = mean_n(hout * nm)
    dhout = np.broadcast_to(dpred[:, None, :] / dtype(N), (B, N, dpred.shape[1])) * nm
    dh = dhout @ sd[p + "embedding_out.weight"]
    dx = np.zeros((B, N, 3), dtype=dtype)
    dd0 = np.zeros((B, N, N, 1), dtype=dtype)
    for l in reversed(range(cfg["n_layers"])):
        gp = f"{p}gcl_{l}."
        L = C["layers"][l]
        W1 = sd[gp + "edge_mlp.0.weight"]
        Wn1 = sd[gp + "node_mlp.0.weight"]
        # h' = (h + Wn2 silu(npre) + bn2) * nm ; x' = (x + sum_j trans) * nm
        dhm = dh * nm
        dxm = dx * nm
        dn1 = dhm @ sd[gp + "node_mlp.2.weight"]
        dnpre = dn1 * _dsilu(L["npre"])
        dnin = dnpre @ Wn1
        dh_prev = dhm + dnin[..., :H]
        dagg = dnin[..., H:]
        de = np.broadcast_to(dagg[:, :, None, :], (B, N, N, H)).copy()
        dtrans = np.broadcast_to(dxm[:, :, None, :], (B, N, N, 3))
        dtau = (dtrans * L["cdiff"]).sum(-1, keepdims=True) * em
        dcdiff = dtrans * L["tau"] * em
        if cfg["tanh"]:
            th = L["tau"] / R
            dphi = dtau * R * (one - th * th)
        else:
            dphi = dtau
        dc1 = dphi * sd[gp + "coord_mlp.2.weight"].reshape(1, 1, 1, H)
        dcpre = dc1 * _dsilu(L["cpre"])
        de = de + dcpre @ sd[gp + "coord_mlp.0.weight"]
        # e = m * a * em
        dm = de * L["a"] * em
        if cfg["attention"]:
            da = (de * L["m"] * em).sum(-1, keepdims=True)
            ds = da * L["a"] * (one - L["a"])
            dm = dm + ds * sd[gp + "att_mlp.0.weight"].reshape(1, 1, 1, H)
        dv = dm * _dsilu(L["v"])
        dt1 = dv @ sd[gp + "edge_mlp.2.weight"]
        du = dt1 * _dsilu(L["u"])
        dinp = du @ W1  # [B,N,N,2H+2]
        dh_prev = dh_prev + dinp[..., :H].sum(axis=2) + dinp[..., H : 2 * H].sum(axis=1)
        dr = dinp[..., 2 * H : 2 * H + 1]
        dd0 = dd0 + dinp[..., 2 * H + 1 : 2 * H + 2]
        # radial / cdiff wrt x (gcl.py:308-316): diff = x_i - x_j, r=|diff|^2, n=sqrt(r+1e-8), cdiff=diff/(n+1)
        x = L["x"]
        diff = x[:, :, None, :] - x[:, None, :, :]
        nrm = np.sqrt(L["radial"] + dtype(1e-8))
        ddiff = dcdiff / (nrm + one) - diff * ((dcdiff * diff).sum(-1, keepdims=True) / ((nrm + one) ** 2 * nrm))
        ddiff = ddiff + dtype(2.0) * diff * dr
        dx_prev = dxm + ddiff.sum(axis=2) - ddiff.sum(axis=1)
        dh, dx = dh_prev, dx_prev
    # embedding + d0 + input masking
    dh0 = dh @ sd[p + "embedding.weight"]
    x0 = C["x0"]
    diff0 = x0[:, :, None, :] - x0[:, None, :, :]
    g0 = dtype(2.0) * diff0 * dd0
    dx = dx + g0.sum(axis=2) - g0.sum(axis=1)
    grad = np.concatenate([dx, dh0[..., :-1]], axis=2) * nm
    return pred, grad


# --------------------------------------------------------------------------------------
# target functions (generation_guidance.py:200-211) expressed as linear functionals of pred
# --------------------------------------------------------------------------------------
def target_max_gap_weights(K):
    """-pred[:,1]  (generation_guidance.py:200-203)."""
    w = np.zeros(K, dtype=F32)
    w[1] = -1.0
    return w


def target_opv_weights(K, std):
    """ip + ea + 3*gap on pred*std+mean (generation_guidance.py:205-211; models_edm.py:186-188).
    d/dpred = [3*std0, 0, std2, std3, ...]."""
    w = np.zeros(K, dtype=F32)
    w[0] = 3.0 * std[0]
    w[2] = std[2]
    w[3] = std[3]
    return w


# --------------------------------------------------------------------------------------
# reverse steps + decode  (en_diffusion.py:807-935, 533-560)
# --------------------------------------------------------------------------------------
def _combined_noise(eps_raw, nm, std=1.0):
    """sample_combined_position_feature_noise (en_diffusion.py:937-956) given raw N(0,1) draws
    eps_raw [B,N,3+F] (x part first, matching the reference's call order)."""
    dt = eps_raw.dtype.type
    ex = eps_raw[:, :, :3] * dt(std) * nm
    ex = remove_mean_with_mask(ex, nm)
    eh = eps_raw[:, :, 3:] * dt(std) * nm
    return np.concatenate([ex, eh], axis=2)


def sample_edm_t(cfg, gamma, x, onehot, t_int, node_mask, eps_raw, dtype=F32):
    """cond_prediction/train_cond_predictor.py:47-61: z_t = alpha_t * normalize([x | h]) + sigma_t * eps with
    gamma looked up at t_int (en_diffusion.py:220-223), normalize = en_diffusion.py:384-392."""
    B, N, _ = x.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    nv = cfg["normalize_factors"]
    xh = np.concatenate([np.asarray(x, dtype) / dtype(nv[0]),
                         (np.asarray(onehot, dtype) - dtype(0.0)) / dtype(nv[1]) * nm], axis=2)
    g = gamma[np.asarray(t_int).astype(np.int64)].astype(dtype).reshape(B, 1, 1)
    alpha = np.sqrt(_sigmoid(-g))
    sigma = np.sqrt(_sigmoid(g))
    return alpha * xh + sigma * _combined_noise(np.asarray(eps_raw, dtype=dtype), nm)


def step_unguided(edm_sd, cfg, gamma, s_idx, z_t, node_mask, edge_mask, eps_raw, dtype=F32):
    """sample_p_zs_given_zt (en_diffusion.py:807-852). s = s_idx/T, t = (s_idx+1)/T."""
    T = cfg["diffusion_steps"]
    B, N, D = z_t.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    z_t = np.asarray(z_t, dtype=dtype)
    c = step_coefficients(gamma, s_idx, s_idx + 1, dtype)
    t_val = dtype(F32(s_idx + 1) / F32(T))
    eps_hat = edm_phi(edm_sd, cfg, z_t, t_val, nm, edge_mask, dtype)
    mu = z_t / c["alpha_ts"] - c["eps_coef"] * eps_hat
    zs = mu + c["sigma"] * _combined_noise(np.asarray(eps_raw, dtype=dtype), nm)
    zs = np.concatenate([remove_mean_with_mask(zs[:, :, :3], nm), zs[:, :, 3:]], axis=2)
    return zs


def step_guided(edm_sd, cfg, pred_sd, pcfg, gamma, s_idx, z_t, node_mask, edge_mask, eps_raw,
                target_w, scale, dtype=F32, return_aux=False, target_z=None):
    """sample_p_zs_given_zt_guidance (en_diffusion.py:854-935).  target_w is either the weight vector of a target
    linear in the predictor outputs, T(pred) = target_w . pred (+const), or a callable
    target_grad(pred [B,K], t) -> dT/dpred [B,K] for an arbitrary target (the chain rule through the closure that
    torch.autograd applies at en_diffusion.py:899-903).  The predictor is evaluated at (z_s, t) -- t, not s
    (en_diffusion.py:902).  target_z (instead of target_w): a callable (z_s, pred, t) -> (dT/dpred [B,K], dT/dz [B,N,D]) for a
    target that also depends on z OUTSIDE the predictor; dT/dz is the direct part (pred held fixed): the total derivative
    autograd takes at en_diffusion.py:900-903 is the predictor path plus it."""
    T = cfg["diffusion_steps"]
    B, N, D = z_t.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    z_t = np.asarray(z_t, dtype=dtype)
    c = step_coefficients(gamma, s_idx, s_idx + 1, dtype)
    t_val = dtype(F32(s_idx + 1) / F32(T))
    eps_hat = edm_phi(edm_sd, cfg, z_t, t_val, nm, edge_mask, dtype)
    eps_hat = np.nan_to_num(eps_hat, nan=0.0, posinf=np.finfo(dtype).max, neginf=np.finfo(dtype).min)
    mu = z_t / c["alpha_ts"] - c["eps_coef"] * eps_hat
    zs = mu + c["sigma"] * _combined_noise(np.asarray(eps_raw, dtype=dtype), nm)
    direct = None
    if target_z is not None:
        pred0 = predictor_forward(pred_sd, pcfg, zs, nm, edge_mask, t_val, dtype)
        gp, gz = target_z(zs, pred0, float(t_val))
        dpred = np.asarray(gp, dtype=dtype).reshape(B, -1) * dtype(scale)
        direct = np.asarray(gz, dtype=dtype).reshape(B, N, D) * dtype(scale) * nm
    elif callable(target_w):
        pred0 = predictor_forward(pred_sd, pcfg, zs, nm, edge_mask, t_val, dtype)
        dpred = np.asarray(target_w(pred0, float(t_val)), dtype=dtype).reshape(B, -1) * dtype(scale)
    else:
        dpred = np.broadcast_to(np.asarray(target_w, dtype=dtype) * dtype(scale), (B, len(target_w)))
    pred, grad = predictor_grad(pred_sd, pcfg, zs, nm, edge_mask, t_val, dpred, dtype)
    if direct is not None:
        grad = grad + direct
    gnorm = np.sqrt((grad.reshape(B, -1) ** 2).sum(-1))
    clip = np.minimum(dtype(10.0) / (gnorm + dtype(1e-6)), dtype(1.0))
    grad = grad * clip[:, None, None]
    grad = np.concatenate([remove_mean_with_mask(grad[:, :, :3], nm), grad[:, :, 3:]], axis=2)
    zs = zs - c["sigma"] * grad
    zs = np.concatenate([remove_mean_with_mask(zs[:, :, :3], nm), zs[:, :, 3:]], axis=2)
    if np.isnan(zs).any():
        zs = np.nan_to_num(zs, nan=0.0, posinf=np.finfo(dtype).max, neginf=np.finfo(dtype).min)
    if return_aux:
        return zs, dict(pred=pred, grad=grad, gnorm=gnorm)
    return zs


def decode_z0(edm_sd, cfg, gamma, z0, node_mask, edge_mask, eps_raw, dtype=F32):
    """sample_p_xh_given_z0 (en_diffusion.py:533-560) + unnormalize (:406-415), include_charges=False.
    Returns x [B,N,3] (un-normalised) and one-hot h [B,N,F]."""
    B, N, D = z0.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    z0 = np.asarray(z0, dtype=dtype)
    g0 = gamma[0].astype(dtype)
    sigma_x = np.exp(dtype(0.5) * g0)  # SNR(-0.5*gamma_0) = exp(0.5*gamma_0)
    eps_hat = edm_phi(edm_sd, cfg, z0, dtype(0.0), nm, edge_mask, dtype)
    sigma_0 = np.sqrt(_sigmoid(g0))
    alpha_0 = np.sqrt(_sigmoid(-g0))
    mu_x = dtype(1.0) / alpha_0 * (z0 - sigma_0 * eps_hat)
    xh = mu_x + sigma_x * _combined_noise(np.asarray(eps_raw, dtype=dtype), nm)
    nv = cfg["normalize_factors"]
    x = xh[:, :, :3] * dtype(nv[0])
    h_cat = (z0[:, :, 3:] * dtype(nv[1]) + dtype(0.0)) * nm
    F = h_cat.shape[2]
    onehot = np.zeros_like(h_cat)
    idx = np.argmax(h_cat, axis=2)
    np.put_along_axis(onehot, idx[:, :, None], 1.0, axis=2)
    return x, onehot * nm


def sample(edm_sd, cfg, node_mask, edge_mask, noise, std=1.0, pred_sd=None, pcfg=None, target_w=None,
           scale=1.0, dtype=F32, keep=None, target_z=None):
    """EnVariationalDiffusion.sample / .sample_guidance (en_diffusion.py:958-1067).

    noise [T+2,B,N,3+F] raw N(0,1): noise[0] -> z_T (scaled by std), noise[1+k] -> k-th reverse
    step (s = T-1-k), noise[T+1] -> final decode.
    """
    T = cfg["diffusion_steps"]
    gamma = gamma_table(cfg["diffusion_noise_schedule"], T, cfg["diffusion_noise_precision"])
    B, N, _ = node_mask.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    z = _combined_noise(np.asarray(noise[0], dtype=dtype), nm, std)
    for k, s in enumerate(reversed(range(T))):
        if target_w is None and target_z is None:
            z = step_unguided(edm_sd, cfg, gamma, s, z, nm, edge_mask, noise[1 + k], dtype)
        else:
            z = step_guided(edm_sd, cfg, pred_sd, pcfg, gamma, s, z, nm, edge_mask, noise[1 + k],
                            target_w, scale, dtype, target_z=target_z)
        if keep is not None:
            keep.append(z.copy())
    x, h = decode_z0(edm_sd, cfg, gamma, z, nm, edge_mask, noise[T + 1], dtype)
    max_cog = np.abs(x.sum(axis=1, keepdims=True)).max()
    if max_cog > 5e-2:  # en_diffusion.py:1000-1006
        x = remove_mean_with_mask(x, nm)
    return x, h, z


def sample_chain(edm_sd, cfg, node_mask, edge_mask, noise, keep_frames, std=1.0, dtype=F32):
    """EnVariationalDiffusion.sample_chain (en_diffusion.py:1118-1174) -> chain [keep_frames,B,N,3+F]
    (the reference returns it viewed as [keep_frames*B, N, 3+F])."""
    T = cfg["diffusion_steps"]
    gamma = gamma_table(cfg["diffusion_noise_schedule"], T, cfg["diffusion_noise_precision"])
    B, N, _ = node_mask.shape
    nm = np.asarray(node_mask, dtype=dtype).reshape(B, N, 1)
    nv = cfg["normalize_factors"]
    z = _combined_noise(np.asarray(noise[0], dtype=dtype), nm, std)
    chain = np.zeros((keep_frames,) + z.shape, dtype=dtype)
    for k, s in enumerate(reversed(range(T))):
        z = step_unguided(edm_sd, cfg, gamma, s, z, nm, edge_mask, noise[1 + k], dtype)
        frame = np.concatenate([z[:, :, :3] * dtype(nv[0]), (z[:, :, 3:] * dtype(nv[1]) + dtype(0.0)) * nm], axis=2)
        chain[(s * keep_frames) // T] = frame  # unnormalize_z (en_diffusion.py:417-431)
    x, h = decode_z0(edm_sd, cfg, gamma, z, nm, edge_mask, noise[T + 1], dtype)
    chain[0] = np.concatenate([x, h], axis=2)
    return chain
