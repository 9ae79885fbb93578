"""GPU tests at BASELINE.json's full sizes: parity against the oracle where the oracle finishes in seconds
(teacher-forced steps at the default architectures), and size-independent properties for the big runs
(E(3) equivariance, permutation equivariance, masking, zero centre of gravity, determinism, sharding)."""
import numpy as np
import pytest

from gaudi_amd import synth
from tests.helpers import max_norm_err, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def O():
    from oracle import gaudi_oracle
    return gaudi_oracle


def _engine(dataset, T=1000, amp=False, guided=True):
    from gaudi_amd.engine import Engine
    F = synth.num_node_features(dataset)
    eargs = synth.edm_args(dataset=dataset, diffusion_steps=T)
    pargs = synth.pred_args(dataset=dataset)
    esd = synth.synth_edm_state_dict(eargs, F, seed=0, amplify_coord=amp)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1, amplify_coord=amp)
    eng = Engine(0)
    eng.load_edm(eargs, esd)
    if guided:
        eng.load_predictor(pargs, psd)
    return eng, eargs, pargs, esd, psd


def _z(nm, F, seed):
    B, N = nm.shape[0], nm.shape[1]
    z = np.random.default_rng(seed).standard_normal((B, N, 3 + F)).astype(np.float32) * nm
    z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1, keepdims=True), 1) * nm
    return z.astype(np.float32)


@pytest.mark.parametrize("dataset,nodes,w", [("cata", [11, 11, 7, 11, 4, 9], [0, -1, 0, 0, 0]),
                                             ("hetro", [10, 3, 7, 5], [3, 0, 1, 1, 0])])
def test_fullsize_guided_step_vs_oracle(O, dataset, nodes, w):
    """Default architectures (EDM nf=192 L=9, predictor nf=196 L=12), one teacher-forced guided step (C3 / C4 shapes)."""
    eng, eargs, pargs, esd, psd = _engine(dataset, amp=True)
    F = synth.num_node_features(dataset)
    nm, em = O.build_masks(nodes, max(nodes), dataset != "cata")
    z = _z(nm, F, 3)
    eps = np.random.default_rng(4).standard_normal(z.shape).astype(np.float32)
    gamma = O.gamma_table("polynomial_2", 1000, 1e-5)
    w = np.asarray(w, np.float32)
    for s in (999, 400, 0):
        got = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
        # amplified coordinate heads + 1/alpha_{t|s} make this step sensitive to rounding: at s = 999 the fp32 oracle itself
        # sits 5.9e-5 (element-wise) from its own float64 evaluation, the kernel 4.4e-5, and the two fp32 results 1.0e-4
        # from each other.  The 1e-4 bar is therefore applied against the float64 evaluation of the same restatement
        # (what both approximate); against the fp32 oracle the two rounding errors add up.
        want64 = O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6, dtype=np.float64)
        assert rel_err(got, want64) < TOL, s
        want = O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6)
        assert rel_err(got, want) < 2 * TOL and max_norm_err(got, want) < TOL, s
        got_u = eng.step(s, z, nm, em, eps)  # same conditioning as the guided step (the guidance term is small here)
        assert rel_err(got_u, O.step_unguided(esd, eargs, gamma, s, z, nm, em, eps, dtype=np.float64)) < TOL, s
        want_u = O.step_unguided(esd, eargs, gamma, s, z, nm, em, eps)
        assert rel_err(got_u, want_u) < 2 * TOL and max_norm_err(got_u, want_u) < TOL, s
    eng.close()


def _rot(seed):
    q, _ = np.linalg.qr(np.random.default_rng(seed).standard_normal((3, 3)))
    return (q * np.sign(np.linalg.det(q))).astype(np.float32)


def test_e3_and_permutation_equivariance_fullsize(O):
    """phi is E(3)-equivariant on x, invariant on h; the predictor is invariant; both commute with node relabelling."""
    eng, eargs, pargs, esd, psd = _engine("hetro", amp=True)
    nodes = [10, 6, 8, 3] * 8  # B=32, N=20
    nm, em = O.build_masks(nodes, 10, True)
    B, N = nm.shape[0], nm.shape[1]
    z = _z(nm, 12, 7)
    t = np.linspace(0.1, 0.9, B).astype(np.float32)
    eps = eng.phi(z, t, nm, em)
    R = _rot(1)
    zr = z.copy()
    zr[:, :, :3] = z[:, :, :3] @ R.T
    eps_r = eng.phi(zr, t, nm, em)
    want = eps.copy()
    want[:, :, :3] = eps[:, :, :3] @ R.T
    assert rel_err(eps_r, want) < 2e-5
    pred = eng.predictor_fwd(z, t, nm, em)
    assert rel_err(eng.predictor_fwd(zr, t, nm, em), pred) < 2e-5
    # permutation of the nodes of every molecule (masks permuted consistently)
    perm = np.random.default_rng(2).permutation(N)
    zp, nmp = z[:, perm], nm[:, perm]
    emp = em.reshape(B, N, N)[:, perm][:, :, perm]
    assert rel_err(eng.phi(zp, t, nmp, emp), eps[:, perm]) < 2e-5
    assert rel_err(eng.predictor_fwd(zp, t, nmp, emp), pred) < 2e-5
    eng.close()


def test_c3_full_chain_properties():
    """BASELINE configs[2]: B=256, N=11, T=1000, gap guidance -- properties the domain offers at full size."""
    eng, *_ = _engine("cata")
    B, N = 256, 11
    nm = np.ones((B, N), np.float32)
    em = np.broadcast_to(1 - np.eye(N, dtype=np.float32), (B, N, N)).copy()
    w = np.array([0, -1, 0, 0, 0], np.float32)
    x, h, d = eng.sample(nm, em, seed=5, target_w=w, scale=0.6)
    assert np.isfinite(x).all() and d["nan_count"] == 0
    assert d["max_masked_leak"] == 0 and d["max_cog_rel"] < 1e-2
    assert np.array_equal(h.sum(-1), nm)  # every live node carries exactly one class
    x2, h2, _ = eng.sample(nm, em, seed=5, target_w=w, scale=0.6)
    assert np.array_equal(x, x2) and np.array_equal(h, h2)  # bitwise reproducible (no atomics anywhere)
    xs, hs, _ = eng.sample(nm[128:], em[128:], seed=5, sample_offset=128, target_w=w, scale=0.6)
    assert np.array_equal(xs, x[128:])  # shard [128,256) alone == rows of the full batch
    x3, _, _ = eng.sample(nm, em, seed=6, target_w=w, scale=0.6)
    assert not np.array_equal(x, x3)
    eng.close()


def test_c4_shape_hetero_mixed_guided(O):
    """BASELINE configs[3] shape (hetero F=12, orientation nodes, mixed 3-10 rings, multi-objective target) at a
    reduced batch/step count; checks masking incl. padded molecules and the padded-ring identity edges."""
    eng, eargs, pargs, esd, psd = _engine("hetro", T=30)
    rings = np.random.default_rng(1).integers(3, 11, size=64)
    nm, em = O.build_masks(rings, 10, True)
    w = O.target_opv_weights(5, np.ones(5, np.float32))
    x, h, d = eng.sample(nm, em, seed=3, target_w=w, scale=0.6)
    assert np.isfinite(x).all() and d["max_masked_leak"] == 0 and d["max_cog_rel"] < 1e-2
    live = nm[:, :, 0] > 0
    assert np.array_equal(h.sum(-1) > 0, live)
    # chain parity against the oracle on a few molecules with the device's own noise stream (host Philox twin).
    # Guided chains with untrained weights amplify rounding noise (the reference's own fp32-vs-fp64 spread is
    # 1.3e-2 at T=50), so the guided chain is held to that spread and the 1e-4 bar is
    # applied to the unguided chain and to the teacher-forced steps above.
    from gaudi_amd.philox import philox_normal
    sel = [0, 1, 2, 3]
    noise = philox_normal(3, 0, 4, 20 * 15, 0, 32).reshape(32, 4, 20, 15)
    em4 = em.reshape(64, 20, 20)[sel]
    xo, ho, _ = O.sample(esd, eargs, nm[sel], em4, noise, pred_sd=psd, pcfg=pargs, target_w=w, scale=0.6)
    assert max_norm_err(x[sel], xo) < 1.3e-2  # the spread is a max-norm figure
    xu, hu, _ = eng.sample(nm[sel], em4, seed=3)
    xou, hou, _ = O.sample(esd, eargs, nm[sel], em4, noise)
    assert rel_err(xu, xou) < TOL and np.array_equal(hu, hou)
    eng.close()
