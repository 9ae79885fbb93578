"""GPU evidence added in round 5: closures that only LOOK affine keep the callback path's result, the unguided full
batch against the C++ port, and the fp16-pair node GEMMs held to the fp32 instruction's accuracy."""
import os
import types

import numpy as np
import pytest

from tests.helpers import TINY, TINY_P, cfg_of, edm_from_cfg, pred_from_cfg

pytestmark = pytest.mark.gpu


def test_closures_that_only_look_affine_follow_the_callback_path(golden):
    """a target closure whose python control flow depends on t -- here guidance with twice the weight
    inside a window of t that the three-point affine probe does not visit -- is recognised as affine at the probed points, runs
    fused, and is caught by the all-t check that the host runs beside the device call: the fused result is discarded and the
    same call (same noise stream) runs through the callback path.  Result = PredTarget on the same closure, bit for bit; a
    closure that is switched OFF at the probed points (w = 0) never reaches the fused kernel at all."""
    import torch

    from gaudi_amd import sampling_edm
    from gaudi_amd.models_edm import PredTarget, get_cond_predictor_model, get_model
    g = golden("g7_end_to_end")
    cfg = cfg_of(g, "hetro_tiny")
    base = dict(dataset=cfg["dataset"], amp=cfg["amp"])
    eargs, esd = edm_from_cfg(dict(base, over=TINY, wseed=cfg["eseed"]), diffusion_steps=cfg["T"])
    pargs, psd = pred_from_cfg(dict(base, over=TINY_P, wseed=cfg["pseed"]))
    model, _, _ = get_model(eargs, state_dict=esd)
    cond_predictor = get_cond_predictor_model(pargs, None, state_dict=psd)
    nodes = [3, 5, 4, 2]
    args = types.SimpleNamespace(device="cuda", dataset=cfg["dataset"], max_nodes=10)
    T = cfg["T"]

    def in_window(t):
        t = float(torch.as_tensor(t).reshape(-1)[0])
        return 0.6 < t < 0.9

    def windowed(_input, _node_mask, _edge_mask, _t):
        pred = cond_predictor(_input, _node_mask, _edge_mask, _t)
        return -2 * pred[:, 1] if in_window(_t) else -pred[:, 1]

    def off_at_the_probes(_input, _node_mask, _edge_mask, _t):
        pred = cond_predictor(_input, _node_mask, _edge_mask, _t)
        return -pred[:, 1] if in_window(_t) else 0 * pred[:, 1]

    def run(target):
        model.seed, model.sample_offset = 11, 0
        model.engine.profile_reset(True)
        x, h, _, _ = sampling_edm.sample_guidance(args, model, target, nodes, scale=0.6)
        n = model.engine.profile_get()[0]
        model.engine.profile_reset(False)
        return x.numpy(), h.numpy(), n, dict(model.last_diag)

    fused = -(-T // 25)
    for closure in (windowed, off_at_the_probes):
        fn = lambda p, t, c=closure: _through(cond_predictor, c, p, t)
        xa, ha, na, da = run(closure)
        xb, hb, nb, _ = run(PredTarget(cond_predictor, fn))
        assert np.array_equal(xa, xb) and np.array_equal(ha, hb)
        if closure is windowed:  # fused chain (discarded) + the callback chain
            assert da.get("affine_recheck_failed") == 1 and na == fused + nb, (na, nb)
        else:
            assert "affine_recheck_failed" not in da and na == nb, (na, nb)
    # the guidance inside the window matters (otherwise the test would pass on an unguided chain)
    x0, _, _, _ = run(lambda z, nm, em, t: -cond_predictor(z, nm, em, t)[:, 1])
    assert not np.array_equal(x0, xa)
    model.engine.close()


def _through(cond_predictor, closure, p, t):
    """closure(z, nm, em, t) with the attached predictor returning p (what GaudiModel._trace_closure does)."""
    import torch
    cond_predictor._override, cond_predictor._override_used = p, False
    try:
        return closure(None, None, None, torch.full((p.shape[0], 1), float(t)))
    finally:
        cond_predictor._override = None


# ------------------------------------------------------------------------------------------------ fp16-pair GEMMs: range
@pytest.mark.parametrize("scale", [3e5, 1e-7, 1.0])
@pytest.mark.parametrize("nf_e,nf_p,n_layers", [(192, 196, 3), (36, 36, 2)])
def test_fp16_pair_gemms_outside_fp16_range_vs_float64(monkeypatch, scale, nf_e, nf_p, n_layers):
    """the node and edge GEMMs run on fp16 pairs (w8_nodes_f16.h, w8_split.h) behind power-of-two scales per
    node / per edge column.  Node features of 3e5 (far above fp16's 65 504: every activation of the first layers overflows an
    unscaled fp16) and of 1e-7 (below fp16's smallest normal number) must come out as close to the oracle's float64 evaluation
    as on the fp32-instruction kernels: denoiser output, predictor output and its input gradient at 1e-4 of the tensor's
    largest entry, and within a small factor of the fp32 kernels' own error."""
    from gaudi_amd import synth
    from gaudi_amd.engine import Engine
    from oracle import gaudi_oracle as O
    from tests.helpers import max_norm_err
    eargs = synth.edm_args(nf=nf_e, n_layers=n_layers, diffusion_steps=100)
    pargs = synth.pred_args(nf=nf_p, n_layers=n_layers)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=21)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=22)
    for sd, key in ((esd, "dynamics.egnn.embedding."), (psd, "egnn.embedding.")):
        sd[key + "weight"] = (np.asarray(sd[key + "weight"]) * np.float32(scale)).astype(np.float32)
        sd[key + "bias"] = (np.asarray(sd[key + "bias"]) * np.float32(scale)).astype(np.float32)
    nm, em = O.build_masks([11, 7, 9, 4], 11, False)
    B, N = nm.shape[0], nm.shape[1]
    rng = np.random.default_rng(6)
    z = rng.standard_normal((B, N, 4)).astype(np.float32) * nm
    z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1, keepdims=True), 1) * nm
    t = np.linspace(0.2, 0.8, B).astype(np.float32)
    w = np.broadcast_to(np.array([0.5, -1.0, 0.25, 0.0, 1.0], np.float32), (B, 5))
    ref_eps = O.edm_phi(esd, eargs, z, t, nm, em, dtype=np.float64)
    ref_pred, ref_grad = O.predictor_grad(psd, pargs, z, nm, em, t, w, dtype=np.float64)
    errs = {}
    for math in ("split", "fp32"):
        monkeypatch.setenv("GAUDI_EDGE_MATH", math)
        eng = Engine(0)
        eng.load_edm(eargs, esd)
        eng.load_predictor(pargs, psd)
        eps = eng.phi(z, t, nm, em)
        pred, grad = eng.predictor_grad(z, t, nm, em, w)
        assert eng.kernel_variant()[1] == 8 and eng.edge_math()[1] == (1 if math == "split" else 0)
        assert np.isfinite(eps).all() and np.isfinite(pred).all() and np.isfinite(grad).all()
        errs[math] = (max_norm_err(eps, ref_eps), max_norm_err(pred, ref_pred), max_norm_err(grad, ref_grad))
        eng.close()
    for es, ef in zip(errs["split"], errs["fp32"]):
        assert es < 1e-4 and es <= 4.0 * ef + 5e-7, errs


def test_weight_sets_the_fp16_images_refuse_run_on_the_fp32_kernels(monkeypatch):
    """Round 6: the refusal rule of the fp16-pair images is as narrow as the images allow and LOUD.  A node
    matrix 1e-5 below the network's largest entry -- which through round 5 sent the whole network to the fp32-instruction kernels,
    silently, at 0.55 x the speed -- now runs on fp16 pairs (the node images carry 22 bits down to 2^-25), within 1e-4 of the
    oracle's float64 evaluation and as close to it as the fp32-instruction kernels; so does an EDGE matrix 1e-4 below.  What the
    images cannot carry still falls back, and says so: a matrix 1e-12 below the others, or an infinite weight (inf - inf = NaN
    where the fp32 product keeps inf), give the fp32-instruction kernels' results bit for bit, a RuntimeWarning at load time and
    gaudi_last_warning."""
    from gaudi_amd import synth
    from gaudi_amd.engine import Engine
    from oracle import gaudi_oracle as O
    from tests.helpers import rel_err
    eargs = synth.edm_args(nf=64, n_layers=2, diffusion_steps=10)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=3)
    nm, em = O.build_masks([5, 7], 7, False)
    z = np.random.default_rng(0).standard_normal((2, 7, 4)).astype(np.float32) * nm
    t = np.array([0.3, 0.6], np.float32)
    monkeypatch.delenv("GAUDI_EDGE_MATH", raising=False)

    def scaled(**factors):
        sd = dict(esd)
        for k, f in factors.items():
            sd[k] = (np.asarray(esd[k]) * np.float32(f)).astype(np.float32)
        return sd

    def phi(sd, math=None):
        if math:
            monkeypatch.setenv("GAUDI_EDGE_MATH", math)
        eng = Engine(0)
        eng.load_edm(eargs, sd)
        got, em_ = eng.phi(z, t, nm, em), eng.edge_math()
        eng.close()
        monkeypatch.delenv("GAUDI_EDGE_MATH", raising=False)
        return got, em_

    wn2, w2 = "dynamics.egnn.e_block_1.gcl_0.node_mlp.2.weight", "dynamics.egnn.e_block_0.gcl_0.edge_mlp.2.weight"
    for factors in ({wn2: 1e-5}, {w2: 1e-4}):
        sd = scaled(**factors)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            got, mode = phi(sd)
        assert mode[0] == 1 and mode[1] != 0, mode  # fp16 pairs, not the fallback
        want64 = O.edm_phi(sd, eargs, z, t, nm, em, dtype=np.float64)
        ref32, _ = phi(sd, "fp32")
        e16, e32 = rel_err(got, want64), rel_err(ref32, want64)
        assert e16 < 1e-4 and e16 < 4 * e32 + 1e-6, (factors, e16, e32)
    inf = dict(esd)
    inf[wn2] = np.array(esd[wn2], copy=True)
    inf[wn2][1, 2] = np.inf
    for sd, why in ((scaled(**{wn2: 1e-12}), "below the largest entry"), (inf, "infinity")):
        with pytest.warns(RuntimeWarning, match=why):
            got, mode = phi(sd)
        assert mode == (1, 0)  # split configured, this weight set refused
        want, _ = phi(sd, "fp32")
        assert np.array_equal(got, want, equal_nan=True)


# ------------------------------------------------------------------------------------------------ the FR kernel instantiation
@pytest.mark.parametrize("widths", ["tiny", "default"])
def test_fr_instantiation_of_the_resident_kernel_is_bit_identical(monkeypatch, widths):
    """Workgroups of more than 16 node slots run the resident full-ring kernel's FR instantiation (kern8s2_*.hip; sampler_kernel.h:
    V8T<1, false, false, true>): its node GEMMs recompute their lane addresses per call instead of reloading them from scratch --
    the same arithmetic in the same order.  Guided and unguided chains and the denoiser of 17..20-node hetero molecules must equal
    the plain instantiation (GAUDI_NO_FR=1) bit for bit, and a teacher-forced guided step must match the oracle."""
    from gaudi_amd.engine import Engine
    from gaudi_amd.sampling_edm import build_masks
    from oracle import gaudi_oracle as O
    from tests.helpers import rel_err
    from gaudi_amd import synth
    dataset, sizes, T = "hetro", [10, 9, 10, 7, 10], 5  # rings: 2 graph nodes per ring -> up to 20 node slots
    F = synth.num_node_features(dataset)
    over_e, over_p = (TINY, TINY_P) if widths == "tiny" else ({}, {})
    eargs = synth.edm_args(dataset=dataset, diffusion_steps=T, **over_e)
    pargs = synth.pred_args(dataset=dataset, **over_p)
    esd = synth.synth_edm_state_dict(eargs, F, seed=11, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=12, amplify_coord=True)
    nm3, em_flat, N = build_masks(sizes, max(sizes), True)
    B, D = len(sizes), 3 + F
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    assert N > 16
    w = np.array([0.5, -1.0, 0.25, 0.0, 1.0], np.float32)
    monkeypatch.delenv("GAUDI_EDGE_MATH", raising=False)
    monkeypatch.delenv("GAUDI_WAVES", raising=False)
    out = {}
    for tag in ("fr", "plain"):
        if tag == "plain":
            monkeypatch.setenv("GAUDI_NO_FR", "1")
        else:
            monkeypatch.delenv("GAUDI_NO_FR", raising=False)
        eng = Engine(0)
        eng.load_edm(eargs, esd)
        eng.load_predictor(pargs, psd)
        res = [eng.sample(nm, em, return_z0=True, seed=9, sample_offset=4, target_w=w, scale=0.6),
               eng.sample(nm, em, return_z0=True, seed=9, sample_offset=4)]
        assert eng.kernel_variant()[1] == 8 and eng.edge_math()[1] == 1, (eng.kernel_variant(), eng.edge_math())  # resident, full ring
        assert eng.last_launch_shape()[1] > 16  # (what selects the FR instantiation: gaudi_hip.hip, launch())
        rng = np.random.default_rng(5)
        z = O._combined_noise(rng.standard_normal((B, N, D)).astype(np.float32), nm[:, :, None])
        eps = rng.standard_normal((B, N, D)).astype(np.float32)
        step = eng.step(2, z, nm, em, eps, target_w=w, scale=0.6)
        phi = eng.phi(z, np.linspace(0.2, 0.8, B).astype(np.float32), nm[:, :, None], em)
        out[tag] = (res, step, phi)
        if tag == "fr":
            gamma = O.gamma_table("polynomial_2", T, 1e-5)
            want = O.step_guided(esd, eargs, psd, pargs, gamma, 2, z, nm[:, :, None], em, eps, w, 0.6)
            assert rel_err(step, want) < 1e-4
        eng.close()
    for ra, rb in zip(out["fr"][0], out["plain"][0]):
        for u, v in zip(ra, rb):
            if isinstance(u, np.ndarray):
                assert np.array_equal(u, v)
    assert np.array_equal(out["fr"][1], out["plain"][1]) and np.array_equal(out["fr"][2], out["plain"][2])
