"""GPU parity of the split-step guidance path (gaudi_sample_cb): arbitrary differentiable targets of the predictor
outputs.  The denoiser, the predictor forward and its reverse pass stay on the device; only dT/dpred [B,K] is
produced by the caller.  Checked against the reference's golden vectors (g10, autograd through the closure), the
numpy oracle and the fused single-launch path."""
import json

import numpy as np
import pytest

from tests.helpers import TINY, TINY_P, cfg_of, edm_from_cfg, nonlinear_target, nonlinear_target_grad, pred_from_cfg, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine(eargs, esd, pargs, psd):
    from gaudi_amd.engine import Engine
    eng = Engine(0)
    eng.load_edm(eargs, esd)
    eng.load_predictor(pargs, psd)
    return eng


@pytest.mark.parametrize("name", ["cata_tiny", "hetro_tiny"])
def test_constant_gradient_callback_equals_fused_path(golden, name):
    """A callback that returns the constant w is the linear target: same kernels, same order -> bit-identical to
    gaudi_sample(target_w=w), and equal to the reference chain."""
    g = golden("g7_end_to_end")
    cfg = cfg_of(g, name)
    base = dict(dataset=cfg["dataset"], amp=cfg["amp"])
    eargs, esd = edm_from_cfg(dict(base, over=TINY, wseed=cfg["eseed"]), diffusion_steps=cfg["T"])
    pargs, psd = pred_from_cfg(dict(base, over=TINY_P, wseed=cfg["pseed"]))
    eng = _engine(eargs, esd, pargs, psd)
    w = np.zeros(5, np.float32)
    w[1] = -1
    nm, em, noise = g[name + "_node_mask"], g[name + "_edge_mask"], g[name + "_noise"]
    calls = []

    def grad(pred, t):
        calls.append(t)
        return np.broadcast_to(w, pred.shape)

    x1, h1, d1, z1 = eng.sample_callback(nm, em, grad, noise=noise, scale=0.6, return_z0=True)
    x0, h0, d0, z0 = eng.sample(nm, em, noise=noise, target_w=w, scale=0.6, return_z0=True)
    assert len(calls) == cfg["T"] and calls[0] == 1.0 and abs(calls[-1] - 1.0 / cfg["T"]) < 1e-7
    assert np.array_equal(z1, z0) and np.array_equal(x1, x0) and np.array_equal(h1, h0)
    assert rel_err(x1, g[name + "_x_guided"]) < TOL
    # Philox noise + sample_offset go through the same plumbing
    xa, _, _ = eng.sample_callback(nm, em, grad, seed=5, sample_offset=3, scale=0.6)
    xb, _, _ = eng.sample(nm, em, seed=5, sample_offset=3, target_w=w, scale=0.6)
    assert np.array_equal(xa, xb)
    eng.close()


def test_nonlinear_target_chain_vs_reference(golden):
    """T=50 chain guided by T = 0.5*log(1+p1^2) + 0.1*tanh(p0)*p3 + t*p2 against the reference's sample_guidance
    (autograd through the closure), via Engine.sample_callback and via the reference-shaped entry points with a
    PredTarget whose gradient comes from torch.autograd on the [B,K] leaf."""
    import types

    import torch

    from gaudi_amd import sampling_edm
    from gaudi_amd.models_edm import PredTarget, get_cond_predictor_model, get_model
    g = golden("g10_nonlinear_target")
    cfg = json.loads(str(g["chain_cfg"]))
    eargs, esd = edm_from_cfg(dict(dataset=cfg["dataset"], over=TINY, wseed=cfg["eseed"], amp=False), diffusion_steps=cfg["T"])
    pargs, psd = pred_from_cfg(dict(dataset=cfg["dataset"], over=TINY_P, wseed=cfg["pseed"], amp=False))
    eng = _engine(eargs, esd, pargs, psd)
    x, h, diag = eng.sample_callback(g["chain_node_mask"], g["chain_edge_mask"], nonlinear_target_grad,
                                     noise=g["chain_noise"], scale=0.6)
    assert rel_err(x, g["chain_x"]) < TOL
    assert np.array_equal(h, g["chain_h"])
    eng.close()

    model, _, _ = get_model(eargs, state_dict=esd)
    pred = get_cond_predictor_model(pargs, model=model, state_dict=psd)
    model.injected_noise = g["chain_noise"]

    def fn(p, t):
        return 0.5 * torch.log1p(p[:, 1] ** 2) + 0.1 * torch.tanh(p[:, 0]) * p[:, 3] + t * p[:, 2]

    target = PredTarget(pred, fn)
    args = types.SimpleNamespace(device="cuda", dataset="hetro", max_nodes=10)
    x2, h2, nm, em = sampling_edm.sample_guidance(args, model, target, cfg["nodes"], scale=0.6)
    assert rel_err(x2.numpy(), g["chain_x"]) < TOL and np.array_equal(h2.numpy(), g["chain_h"])
    # the target object is also callable with the reference closure signature (get_target_function_values path)
    zt = np.concatenate([x2.numpy(), h2.numpy()], axis=2).astype(np.float32)
    val = target(torch.from_numpy(zt), nm, em, torch.zeros(len(cfg["nodes"]), 1))
    p_np = model.engine.predictor_fwd(zt, 0.0, nm.numpy().reshape(len(cfg["nodes"]), -1),
                                      em.numpy().reshape(len(cfg["nodes"]), zt.shape[1], zt.shape[1]))
    np.testing.assert_allclose(val.numpy(), nonlinear_target(p_np, 0.0)[0], rtol=1e-5, atol=1e-6)
    model.engine.close()


def test_nonlinear_target_full_size_vs_oracle():
    """Default architectures (EDM nf=192 x 9 blocks, predictor nf=196 x 12 layers), hetero masks, short chain:
    the split-step path against the numpy oracle with the same dT/dpred."""
    from oracle import gaudi_oracle as O
    from gaudi_amd import synth
    T = 6
    F = synth.num_node_features("hetro")
    eargs = synth.edm_args(dataset="hetro", diffusion_steps=T)
    esd = synth.synth_edm_state_dict(eargs, F, seed=41)
    pargs = synth.pred_args(dataset="hetro")
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=42)
    nodes = [5, 9, 10, 2]
    nm, em = O.build_masks(nodes, 10, True)
    B, N = nm.shape[0], nm.shape[1]
    rng = np.random.default_rng(7)
    noise = rng.standard_normal((T + 2, B, N, 3 + F)).astype(np.float32)
    eng = _engine(eargs, esd, pargs, psd)
    x, h, diag, z0 = eng.sample_callback(nm.reshape(B, N), em.reshape(B, N, N), nonlinear_target_grad, noise=noise,
                                         scale=0.6, return_z0=True)
    xo, ho, zo = O.sample(esd, eargs, nm, em.reshape(B, N, N), noise, std=1.0, pred_sd=psd, pcfg=pargs,
                          target_w=nonlinear_target_grad, scale=0.6)
    assert rel_err(z0, zo) < TOL
    assert rel_err(x, xo) < TOL
    assert np.array_equal(h, ho)
    eng.close()


def test_callback_errors_surface():
    from oracle import gaudi_oracle as O
    from gaudi_amd import synth
    from gaudi_amd._lib import GaudiError
    eargs = synth.edm_args(diffusion_steps=3, **TINY)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=1)
    pargs = synth.pred_args(**TINY_P)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=2)
    from gaudi_amd.engine import Engine
    nm, em = O.build_masks([3, 4], 4, False)
    nm, em = nm.reshape(2, 4), em.reshape(2, 4, 4)
    eng = Engine(0)
    eng.load_edm(eargs, esd)
    with pytest.raises(GaudiError, match="predictor"):
        eng.sample_callback(nm, em, lambda p, t: np.zeros_like(p))
    eng.load_predictor(pargs, psd)

    def boom(pred, t):
        raise ValueError("target exploded")

    with pytest.raises(ValueError, match="target exploded"):
        eng.sample_callback(nm, em, boom)
    with pytest.raises(GaudiError, match=r"\[B,K\]"):
        eng.sample_callback(nm, em, lambda p, t: np.zeros(3, np.float32))
    # the handle is still usable afterwards
    x, h, d = eng.sample_callback(nm, em, lambda p, t: np.zeros_like(p), seed=1)
    x0, h0, d0 = eng.sample(nm, em, seed=1, target_w=np.zeros(5, np.float32), scale=1.0)
    assert np.array_equal(x, x0)
    eng.close()
