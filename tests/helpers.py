"""Shared helpers for the parity tests: rebuild the synthetic weights a fixture was made with."""
import json

import numpy as np

from gaudi_amd import synth

TINY = dict(nf=32, n_layers=2)
TINY_P = dict(nf=36, n_layers=3)


def max_norm_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_err(a, b, atol_frac=0.1):
    """Element-wise error: the smallest tol for which np.allclose(a, b, rtol=tol, atol=atol_frac*tol*max|b|) holds, i.e.
    `elem_err < 1e-4` is allclose(rtol=1e-4, atol=1e-5*max|b|): a small component may not hide behind a large one."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    scale = max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) / (np.abs(b) + atol_frac * scale)).max())


def rel_err(a, b):
    """The parity metric of every test: max(max-norm relative error, element-wise error)."""
    return max(max_norm_err(a, b), elem_err(a, b))


def cfg_of(fix, name):
    return json.loads(str(fix[name + "_cfg"]))


def edm_from_cfg(cfg, **extra):
    over = dict(cfg.get("over", {}))
    over.update(extra)
    args = synth.edm_args(dataset=cfg["dataset"], **over)
    F = synth.num_node_features(cfg["dataset"])
    sd = synth.synth_edm_state_dict(args, F, seed=cfg.get("wseed", cfg.get("eseed")), amplify_coord=cfg.get("amp", False))
    return args, sd


def pred_from_cfg(cfg, K=5, **extra):
    over = dict(cfg.get("over", {}))
    over.update(extra)
    args = synth.pred_args(dataset=cfg["dataset"], **over)
    F = synth.num_node_features(cfg["dataset"])
    sd = synth.synth_predictor_state_dict(args, F, K, seed=cfg.get("wseed", cfg.get("pseed")), amplify_coord=cfg.get("amp", False))
    return args, sd


def nonlinear_target(pred, t):
    """numpy twin of tools/make_golden.py:nonlinear_target_torch -> (T [B], dT/dpred [B,K]):
    T = 0.5*log(1+p1^2) + 0.1*tanh(p0)*p3 + t*p2  (smooth for the large |pred| of synthetic-weight chains)."""
    pred = np.asarray(pred)
    t = pred.dtype.type(t)
    th = np.tanh(pred[:, 0])
    val = 0.5 * np.log1p(pred[:, 1] ** 2) + 0.1 * th * pred[:, 3] + t * pred[:, 2]
    g = np.zeros_like(pred)
    g[:, 0] = 0.1 * (1.0 - th * th) * pred[:, 3]
    g[:, 1] = pred[:, 1] / (1.0 + pred[:, 1] ** 2)
    g[:, 2] = t
    g[:, 3] = 0.1 * th
    return val, g


def nonlinear_target_grad(pred, t):
    return nonlinear_target(pred, t)[1]


def direct_z_target_grad(node_mask):
    """numpy twin of tools/make_golden.py:direct_z_target_torch, as the (z_s, pred, t) -> (dT/dpred, dT/dz direct) callable the
    oracle's target_z and Engine.sample_callback(with_z=True) take:
    T = -pred[:, 1] + 0.05 * sum_live |x_n|^2 + 0.02 * sum_live z[:, :, 3]."""
    nm = np.asarray(node_mask, np.float32).reshape(node_mask.shape[0], node_mask.shape[1], 1)

    def grad(z, pred, t):
        gp = np.zeros_like(pred)
        gp[:, 1] = -1.0
        gz = np.zeros_like(z)
        gz[:, :, :3] = 0.1 * z[:, :, :3] * nm
        gz[:, :, 3] = 0.02 * nm[:, :, 0]
        return gp, gz

    return grad


def rng_noise(seed, shape):
    """Same stream as tools/make_golden.py:rng_noise (fixtures that store a seed + checksum instead of the draws)."""
    return np.random.Generator(np.random.Philox(key=seed)).standard_normal(shape).astype(np.float32)


def noise_from_fixture(g, shape):
    noise = rng_noise(int(g["noise_seed"]), shape)
    T = shape[0] - 2
    chk = np.array([noise.astype(np.float64).sum(), np.abs(noise).astype(np.float64).sum(), noise[17, 3, 5, 2],
                    noise[T + 1, 7, 10, 3]])
    assert np.array_equal(chk, g["noise_checksum"]), "numpy's Philox/normal stream differs from the one the fixture was made with"
    return noise
