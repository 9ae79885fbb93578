"""GPU parity of the predictor-evaluation path: forward noising fused into the predictor-forward
launch (gaudi_predict_noised) against the reference's sample_edm_t / compute_loss outputs (g13) and the oracle."""
import json

import numpy as np
import pytest

from tests.helpers import TINY, TINY_P, edm_from_cfg, pred_from_cfg, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _models(cfg):
    from gaudi_amd.models_edm import get_cond_predictor_model, get_model
    eargs, esd = edm_from_cfg(dict(dataset=cfg["dataset"], over=TINY, wseed=cfg["eseed"], amp=False))
    pargs, psd = pred_from_cfg(dict(dataset=cfg["dataset"], over=TINY_P, wseed=cfg["pseed"], amp=True))
    model, _, _ = get_model(eargs, state_dict=esd)
    pred = get_cond_predictor_model(pargs, model=model, state_dict=psd)
    return model, pred, eargs


@pytest.mark.parametrize("name", ["cata", "hetro"])
def test_sample_edm_t_and_compute_loss_vs_reference(golden, name):
    import types
    from gaudi_amd import cond_prediction as cp
    g = golden("g13_noised_predictor")
    cfg = json.loads(str(g[name + "_cfg"]))
    model, pred, eargs = _models(cfg)
    x, h, nm, em, y = (g[f"{name}_{k}"] for k in ("x", "h", "node_mask", "edge_mask", "y"))
    T = cfg["T"]
    for tag in ("t0", "t500", "tT", "tmix"):
        ti = g[f"{name}_{tag}_t_int"]
        t = (ti / np.float32(T)).astype(np.float32).reshape(-1, 1)
        zt = cp.sample_edm_t(x, h, model, t, nm, noise=g[f"{name}_{tag}_eps"])
        assert rel_err(zt.numpy(), g[f"{name}_{tag}_zt"]) < 1e-6, tag
        zt2, p = model.engine.predict_noised(x, h, ti, nm, em, noise=g[f"{name}_{tag}_eps"])
        assert np.array_equal(zt2, zt.numpy())
        assert rel_err(p, g[f"{name}_{tag}_pred"]) < TOL, tag
    loss, err = cp.compute_loss(pred, x, h, nm, em, y, model, types.SimpleNamespace(diffusion_steps=T), t_fix=500,
                                noise=g[f"{name}_t500_eps"])
    assert rel_err(err.numpy(), g[f"{name}_t500_err"]) < TOL and abs(float(loss) - g[f"{name}_t500_loss"]) < 1e-4
    model.engine.close()


def test_philox_noise_replay_and_val_epoch():
    """Production noise (device Philox) replayed on the host twin -> oracle; val_epoch / t_sweep over a synthetic loader."""
    import types
    from oracle import gaudi_oracle as O
    from gaudi_amd import cond_prediction as cp
    from gaudi_amd.philox import philox_normal
    cfg = dict(dataset="cata", eseed=21, pseed=22)
    model, pred, eargs = _models(cfg)
    T = model.T
    rng = np.random.default_rng(5)
    nm, em = O.build_masks([4, 11, 7, 9], 11, False)
    B, N = 4, 11
    nm2 = nm.reshape(B, N)
    x = rng.standard_normal((B, N, 3)).astype(np.float32) * nm
    x = x - x.sum(1, keepdims=True) / nm.sum(1, keepdims=True) * nm
    h = nm.copy()  # cata: one ring type
    y = rng.standard_normal((B, 5)).astype(np.float32)
    ti = np.array([0, 137, 900, T], np.int32)
    model.seed, model.sample_offset = 9, 100
    zt, p = model.engine.predict_noised(x, h, ti, nm2, em.reshape(B, N, N), seed=9, sample_offset=100)
    eps = philox_normal(9, 100, B, N * 4, 0, 1)[0].reshape(B, N, 4)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    zo = O.sample_edm_t(eargs, gamma, x, h, ti, nm, eps)
    assert rel_err(zt, zo) < 1e-5
    pargs, psd = pred_from_cfg(dict(dataset="cata", over=TINY_P, wseed=22, amp=True))
    po = O.predictor_forward(psd, pargs, zo, nm, em.reshape(B, N, N), (ti / np.float32(T)).astype(np.float32))
    assert rel_err(p, po) < TOL

    class DS:
        std = np.array([1.5, 0.7, 2.0, 0.9, 1.1], np.float32)

    class Loader(list):
        dataset = DS()

    loader = Loader([(x + 0.3, nm2, em, h, y), (x, nm2, em, h, y)])  # first batch is not mean-free: val_epoch centres it
    mae = cp.val_epoch("test", pred, model, loader, None, None, t_fix=137)
    model.sample_offset -= 2 * B  # replay the same Philox offsets on the oracle
    errs = []
    for k in range(2):
        eps = philox_normal(9, model.sample_offset + k * B, B, N * 4, 0, 1)[0].reshape(B, N, 4)
        z = O.sample_edm_t(eargs, gamma, x, h, np.full(B, 137), nm, eps)
        errs.append(np.abs(O.predictor_forward(psd, pargs, z, nm, em.reshape(B, N, N), np.float32(137 / T)) - y))
    assert abs(mae - float((np.concatenate(errs) * DS.std[None]).mean())) < 1e-4 * max(1.0, abs(mae))
    times, maes = cp.t_sweep(pred, model, loader, None, None, n_points=3)
    assert list(times) == [0.0, T / 2, float(T)] and len(maes) == 3 and np.isfinite(maes).all()
    with pytest.raises(Exception, match="t_int"):
        model.engine.predict_noised(x, h, np.full(B, T + 1), nm2, em.reshape(B, N, N))
    model.engine.close()
