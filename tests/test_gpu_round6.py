"""GPU evidence added in round 6: the per-molecule kernel family of gaudi_sample, V8G launches with every
molecule's nodes compacted, the fp16-image refusal made loud (item 6)."""
import os
import warnings

import numpy as np
import pytest

from gaudi_amd import synth
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _engine(eargs, esd, pargs=None, psd=None, **env):
    from gaudi_amd.engine import Engine
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        eng = Engine(0)  # the knobs are read once, by gaudi_create
    finally:
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    eng.load_edm(eargs, esd)
    if pargs is not None:
        eng.load_predictor(pargs, psd)
    return eng


def _hetero_batch(rings, pad, T, seeds=(21, 22)):
    from gaudi_amd.sampling_edm import build_masks
    F = synth.num_node_features("hetro")
    eargs, pargs = synth.edm_args(dataset="hetro", diffusion_steps=T), synth.pred_args(dataset="hetro")
    esd = synth.synth_edm_state_dict(eargs, F, seed=seeds[0])
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=seeds[1])
    nm3, em_flat, N = build_masks(rings, pad, True)
    B = len(rings)
    return eargs, esd, pargs, psd, F, nm3.reshape(B, N), em_flat.reshape(B, N, N), N


@pytest.mark.parametrize("guided", [True, False])
def test_family_split_a_molecules_kernel_depends_on_its_own_size_only(guided):
    """BASELINE config 4 read literally mixes 6-20 rings (12-40 graph nodes) in one call (sampling_edm.py:172-209 pads to the batch
    maximum, no size cap).  The padded N = 40 is beyond the resident kernels' LDS limit, and through round 5 EVERY molecule of such
    a call ran on the V8G kernels (node buffers in a global scratch).  Round 6: the molecules that fit the resident kernels on
    their own run there, packed, the rest on V8G -- two buckets of one gaudi_sample call.  The bucket of a molecule is a function
    of its own graph: the molecule sampled ALONE (same padding, its own global index), in any shard of the batch, or in the
    whole batch gives the same bits; noise is keyed by the index in the request, so the whole call equals the one-family call
    (the default: the split is opt-in, GAUDI_FAMILY_SPLIT=1 -- as two launches per 25 steps it measured slower) to the two families' agreement (5e-6 per step; 1e-4 over this short chain)."""
    T = 6
    rings = [3, 20, 6, 12, 9, 16, 4, 11]
    eargs, esd, pargs, psd, F, nm, em, N = _hetero_batch(rings, 20, T)
    assert N == 40
    w = np.array([3, 0, 1, 1, 0], np.float32) if guided else None
    B = len(rings)
    eng = _engine(eargs, esd, pargs, psd, GAUDI_FAMILY_SPLIT=1)
    x, h, d = eng.sample(nm, em, seed=5, target_w=w, scale=0.6)
    nres = d["family_split_resident"]
    assert 0 < nres < B, nres
    assert d["max_masked_leak"] == 0 and np.isfinite(x).all()
    x2, h2, _ = eng.sample(nm, em, seed=5, target_w=w, scale=0.6)
    assert np.array_equal(x, x2) and np.array_equal(h, h2)  # bitwise reproducible
    # every molecule alone, with its own global sample index
    for b in range(B):
        xb, hb, db = eng.sample(nm[b:b + 1], em[b:b + 1], seed=5, sample_offset=b, target_w=w, scale=0.6)
        assert np.array_equal(xb[0], x[b]) and np.array_equal(hb[0], h[b]), (b, rings[b], db["family_split_resident"])
    # two shards
    xa, ha, _ = eng.sample(nm[:3], em[:3], seed=5, sample_offset=0, target_w=w, scale=0.6)
    xc, hc, _ = eng.sample(nm[3:], em[3:], seed=5, sample_offset=3, target_w=w, scale=0.6)
    assert np.array_equal(np.concatenate([xa, xc]), x) and np.array_equal(np.concatenate([ha, hc]), h)
    eng.close()
    # one family per call (round 5): same noise stream, the two kernel families agree to a few 1e-6 per step
    eng1 = _engine(eargs, esd, pargs, psd)  # the default
    x1, h1, d1 = eng1.sample(nm, em, seed=5, target_w=w, scale=0.6)
    assert d1["family_split_resident"] == 0 and eng1.node_buffers_global()
    eng1.close()
    assert rel_err(x, x1) < 1e-4 and np.array_equal(h, h1)


def test_family_split_with_injected_noise_vs_oracle():
    """The same call with injected raw draws (the parity path: [T+2][B][N][D], gathered per bucket) against the numpy oracle's
    guided chain: both buckets at 1e-4."""
    from oracle import gaudi_oracle as O
    T = 4
    rings = [4, 20, 7, 13]
    eargs, esd, pargs, psd, F, nm, em, N = _hetero_batch(rings, 20, T, seeds=(31, 32))
    B = len(rings)
    noise = np.random.default_rng(3).standard_normal((T + 2, B, N, 3 + F)).astype(np.float32)
    w = np.array([3, 0, 1, 1, 0], np.float32)
    eng = _engine(eargs, esd, pargs, psd, GAUDI_FAMILY_SPLIT=1)
    x, h, d = eng.sample(nm, em, noise=noise, target_w=w, scale=0.6)
    assert d["family_split_resident"] == 2
    eng.close()
    xo, ho, _ = O.sample(esd, eargs, nm[:, :, None], em, noise, pred_sd=psd, pcfg=pargs, target_w=w, scale=0.6)
    assert rel_err(x, xo) < 1e-4 and np.array_equal(h, ho)


def test_v8g_compacted_nodes_equal_nodes_in_place_bit_for_bit():
    """V8G sampling launches (round 6) place a molecule's live nodes in the FRONT slots of its workgroup (a hetero molecule's rings
    and orientation nodes are two blocks of the padded index range: fewer node-GEMM column tiles when compacted).  Node columns
    are independent, every sum visits its terms in the molecule's own order: the result must not change by a bit against the
    round-5 launch (GAUDI_GN8_PACK=0), guided and unguided."""
    T = 5
    rings = [20, 13, 9, 17]
    eargs, esd, pargs, psd, F, nm, em, N = _hetero_batch(rings, 20, T, seeds=(41, 42))
    w = np.array([3, 0, 1, 1, 0], np.float32)
    outs = []
    for env in ({"GAUDI_FAMILY_SPLIT": 0, "GAUDI_GN8_PACK": 0}, {"GAUDI_FAMILY_SPLIT": 0}):
        eng = _engine(eargs, esd, pargs, psd, **env)
        a = eng.sample(nm, em, seed=9, target_w=w, scale=0.6)
        assert eng.node_buffers_global()
        b = eng.sample(nm, em, seed=9)
        outs.append((a[0], a[1], b[0], b[1]))
        eng.close()
    for u, v in zip(*outs):
        assert np.array_equal(u, v)


def test_refused_fp16_images_warn_and_report():
    """a weight set the fp16-pair images cannot carry (here: an infinite weight) runs the
    fp32-instruction kernels -- about 0.55 x the speed -- and used to do so silently.  Now the load warns, gaudi_last_warning
    carries the reason and every sample() reports it in diag['edge_math_fallback']; a healthy weight set reports None."""
    from oracle import gaudi_oracle as O
    from tests.helpers import TINY, TINY_P
    T = 4
    eargs, pargs = synth.edm_args(diffusion_steps=T, **TINY), synth.pred_args(**TINY_P)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=3)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=4)
    nm, em = O.build_masks([5, 7, 3], 7, False)
    w = O.target_max_gap_weights(5)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng = _engine(eargs, esd, pargs, psd)  # no warning
    x, h, d = eng.sample(nm, em, seed=1, target_w=w, scale=0.6)
    assert d["edge_math_fallback"] is None and eng.edge_math()[1] != 0
    eng.close()
    bad = {k: np.array(v, copy=True) for k, v in psd.items()}
    key = next(k for k in bad if k.endswith("node_mlp.0.weight"))
    bad[key][0, 0] = np.inf
    with pytest.warns(RuntimeWarning, match="fp16-pair images cannot carry"):
        eng = _engine(eargs, esd, pargs, bad)
    x, h, d = eng.sample(nm, em, seed=1, target_w=w, scale=0.6)
    assert "infinity" in d["edge_math_fallback"] and eng.edge_math()[1] == 0
    eng.close()


@pytest.mark.parametrize("widths", ["tiny", "default"])
def test_kept_split_copy_of_h_changes_nothing(widths):
    """Round 6 (VERDICT r5 item 1 ii): where 160 KiB leave the room (C2 / C3: 46 KB free) the fp16-pair split copy of h is KEPT --
    P splits it, Q and the node MLP's first Linear read the same copy, and in the denoiser the next block's P and Q too (the
    EquivariantUpdate leaves h alone, egnn_new.py:119-155): 2 of the denoiser's 5 split passes per block and 1 of the predictor's
    4 per layer go away.  The copy is a function of h alone: guided and unguided chains, a teacher-forced step and the unit entry
    points must not change by a bit against GAUDI_KEEP_H=0, for one molecule per workgroup and for packed small molecules."""
    from oracle import gaudi_oracle as O
    from tests.helpers import TINY, TINY_P
    T = 7
    over_e, over_p = (TINY, TINY_P) if widths == "tiny" else ({}, {})
    eargs, pargs = synth.edm_args(diffusion_steps=T, **over_e), synth.pred_args(**over_p)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=13, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=14, amplify_coord=True)
    nm, em = O.build_masks([11, 11, 4, 5, 7, 11], 11, False)
    B, N = nm.shape[0], nm.shape[1]
    rng = np.random.default_rng(2)
    z = O._combined_noise(rng.standard_normal((B, N, 4)).astype(np.float32), nm)
    eps = rng.standard_normal((B, N, 4)).astype(np.float32)
    w = O.target_max_gap_weights(5)
    outs, kept = [], []
    for env in ({"GAUDI_KEEP_H": 0}, {}):
        eng = _engine(eargs, esd, pargs, psd, **env)
        a = eng.sample(nm, em, seed=3, target_w=w, scale=0.6)
        kept.append(eng.keep_h())
        b = eng.sample(nm, em, seed=3)
        c = eng.step(3, z, nm, em, eps, target_w=w, scale=0.6)
        outs.append((a[0], a[1], b[0], b[1], c))
        eng.close()
    assert kept[0] == 0 and kept[1] > 0, kept
    for u, v in zip(*outs):
        assert np.array_equal(u, v)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    assert rel_err(outs[1][4], O.step_guided(esd, eargs, psd, pargs, gamma, 3, z, nm, em, eps, w, 0.6)) < 1e-4


def test_half_ring_mode_is_reached_and_checked():
    """since the fp16-pair ring shrank to 52 KiB the half-ring mode (two trips per K chunk, SplitGeo<HP, 2>) is only
    reached where 22 node slots leave no room for the full ring -- wide groups (two 11-ring cata molecules per workgroup) -- and
    no test asserted that it still IS reached.  Forced wide groups at the default widths: edge_math reports mode 2, the result
    equals the one-molecule-per-workgroup launch (full ring) bit for bit (bench.py gates the same launch shape at 1 024 molecules
    against the C++ port: secondary.c3_b1024).  Later in round 6 such a group runs on the FULL ring by default, with the predictor's
    fifth node buffer in the workgroup's global scratch (kern8mp_fused.hip; node_buffers_form() = 3; +5 % at 1 024 molecules):
    GAUDI_WIDE_FULL=0 is the half-ring launch; all three give the same bits."""
    from oracle import gaudi_oracle as O
    T = 6
    eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
    esd = synth.synth_edm_state_dict(eargs, 1, seed=51)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=52)
    nm, em = O.build_masks([11, 11, 11, 9], 11, False)
    w = O.target_max_gap_weights(5)
    res = []
    for env, want, form in (({"GAUDI_PAIRS": 0}, 1, 0), ({"GAUDI_PAIRS": 2, "GAUDI_WIDE_FULL": 0}, 2, 0), ({"GAUDI_PAIRS": 2}, 1, 3)):
        eng = _engine(eargs, esd, pargs, psd, **env)
        x, h, _ = eng.sample(nm, em, seed=4, target_w=w, scale=0.6)
        assert eng.edge_math()[1] == want and eng.node_buffers_form() == form, (env, eng.edge_math(), eng.node_buffers_form())
        assert not eng.node_buffers_global()
        wg, slots = eng.last_launch_shape()
        assert (wg, slots) == ((4, 11) if env["GAUDI_PAIRS"] == 0 else (2, 22))
        res.append((x, h))
        eng.close()
    for x, h in res[1:]:
        assert np.array_equal(res[0][0], x) and np.array_equal(res[0][1], h)


def test_wide_groups_on_the_full_ring_at_the_test_widths_and_unguided():
    """The same at the test widths (32, 48) on a ragged hetero batch with the nonlinear-free OPV-like weights, a guided chain and a
    teacher-forced step against the oracle; an UNGUIDED wide launch has no predictor and stays what it was (no global buffer)."""
    from oracle import gaudi_oracle as O
    from tests.helpers import TINY, TINY_P
    T = 6
    eargs, pargs = synth.edm_args(diffusion_steps=T, **TINY), synth.pred_args(**TINY_P)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=53, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=54, amplify_coord=True)
    nm, em = O.build_masks([11, 11, 10, 11, 9, 11], 11, False)
    w = np.array([3, 0, 1, 1, 0], np.float32)
    outs = []
    for env in ({"GAUDI_PAIRS": 2, "GAUDI_WIDE_FULL": 0}, {"GAUDI_PAIRS": 2}):
        eng = _engine(eargs, esd, pargs, psd, **env)
        x, h, _ = eng.sample(nm, em, seed=9, target_w=w, scale=0.6)
        form = eng.node_buffers_form()
        xu, hu, _ = eng.sample(nm, em, seed=9)
        assert eng.node_buffers_form() == 0
        outs.append((x, h, xu, hu, form, eng.edge_math()[1]))
        eng.close()
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][:4], outs[1][:4]))
    # (at the test widths five buffers fit beside the full ring: both launches are the resident full-ring MR kernel)
    assert outs[0][4] == 0 and outs[1][4] in (0, 3)


# ------------------------------------------------------------------------------------------------ sin_embedding denoisers (g22)
@pytest.mark.parametrize("name", ["cata_tiny", "hetro_tiny", "cata_default"])
def test_sin_embedding_phi_vs_reference(golden, name):
    """`--sin_embedding True` (utils/args_edm.py:33; egnn_new.py:269-273,378-391): through round 5 such a checkpoint was refused at
    load.  The 4-wave kernels carry it (24 sinusoid edge features per first Linear, kernse_*.hip).  phi against the REFERENCE's
    fp32 AND float64 outputs: the embedding multiplies sqrt(r) by up to 429 before sin / cos, so fp32 evaluations of the same
    network differ among themselves by the reference's own fp32-vs-fp64 spread (3e-3 on the amplified tiny case) -- the bound is
    1e-4 where that spread allows it and the spread itself where it does not."""
    import json as _json
    from tests.helpers import edm_from_cfg
    g = golden("g22_sin_embedding")
    cfg = _json.loads(str(g[name + "_cfg"]))
    eargs, esd = edm_from_cfg(cfg, diffusion_steps=cfg["T"])
    z, t, nm, em = g[name + "_z"], g[name + "_t"][:, 0], g[name + "_node_mask"], g[name + "_edge_mask"]
    B, N = z.shape[:2]
    eng = _engine(eargs, esd)
    eps = eng.phi(z, t, nm.reshape(B, N), em.reshape(B, N, N))
    assert eng.kernel_variant()[1] == 4  # an 8-wave handle: the call fell to the only family that has the kernels
    spread = rel_err(g[name + "_eps"], g[name + "_eps64"])
    assert rel_err(eps, g[name + "_eps64"]) < max(1e-4, 2 * spread), (rel_err(eps, g[name + "_eps64"]), spread)
    assert rel_err(eps, g[name + "_eps"]) < max(1e-4, 2 * spread)
    assert np.abs(eps * (1 - nm.reshape(B, N, 1))).max() == 0
    # forced onto the kernels with the node buffers in global memory (what a large molecule takes): the same source, a few
    # multiply-adds contracted differently -- last-bit differences, which the embedding amplifies as it does the reference's own
    eng_g = _engine(eargs, esd, GAUDI_FORCE_GN=1)
    eps_g = eng_g.phi(z, t, nm.reshape(B, N), em.reshape(B, N, N))
    assert eng_g.node_buffers_global() and rel_err(eps_g, eps) < max(1e-5, 0.1 * spread), (rel_err(eps_g, eps), spread)
    assert rel_err(eps_g, g[name + "_eps64"]) < max(1e-4, 2 * spread)
    eng.close()
    eng_g.close()


def test_sin_embedding_steps_and_chain(golden):
    """Default widths, sin_embedding denoiser + the ordinary predictor: the reference's teacher-forced unguided and guided steps
    (s = 999, 400, 0) at 1e-4, then a short guided chain through gaudi_sample (finite, masked, reproducible)."""
    import json as _json
    from tests.helpers import edm_from_cfg, pred_from_cfg
    g = golden("g22_sin_embedding")
    name = "cata_default"
    cfg = _json.loads(str(g[name + "_cfg"]))
    eargs, esd = edm_from_cfg(cfg, diffusion_steps=cfg["T"])
    pargs, psd = pred_from_cfg(dict(dataset=cfg["dataset"], wseed=cfg["pseed"]))
    z, nm, em, eps = g[name + "_z"], g[name + "_node_mask"], g[name + "_edge_mask"], g[name + "_step_noise"]
    B, N = z.shape[:2]
    nm2, em3 = nm.reshape(B, N), em.reshape(B, N, N)
    w = np.array([0, -1, 0, 0, 0], np.float32)
    eng = _engine(eargs, esd, pargs, psd)
    for s in (999, 400, 0):
        assert rel_err(eng.step(s, z, nm2, em3, eps), g[f"{name}_zs_unguided_s{s}"]) < 1e-4, s
        assert rel_err(eng.step(s, z, nm2, em3, eps, target_w=w, scale=0.6), g[f"{name}_zs_guided_s{s}"]) < 1e-4, s
        assert eng.kernel_variant()[1] == 4
    eng.close()
    eargs6 = dict(eargs, diffusion_steps=6)
    eng = _engine(eargs6, esd, pargs, psd)
    x, h, d = eng.sample(nm2, em3, seed=3, target_w=w, scale=0.6)
    x2, h2, _ = eng.sample(nm2, em3, seed=3, target_w=w, scale=0.6)
    assert np.isfinite(x).all() and d["nan_count"] == 0 and np.array_equal(x, x2) and np.array_equal(h, h2)
    assert np.abs(x * (1 - nm2[:, :, None])).max() == 0
    eng.close()


def test_sin_embedding_large_molecule_vs_oracle():
    """A sin_embedding denoiser on a molecule beyond the LDS limit (hetero 20 rings = 40 graph nodes, default widths): the kernels
    with the node buffers in global memory, the guided step as two launches (denoiser, then predictor) -- phi and a guided
    teacher-forced step against the oracle (pinned on the reference by g22) at 1e-4."""
    from oracle import gaudi_oracle as O
    from gaudi_amd.sampling_edm import build_masks
    T, s = 1000, 400
    F = synth.num_node_features("hetro")
    eargs, pargs = synth.edm_args(dataset="hetro", diffusion_steps=T, sin_embedding=True), synth.pred_args(dataset="hetro")
    esd = synth.synth_edm_state_dict(eargs, F, seed=61)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=62)
    rings = [20, 13]
    nm3, em_flat, N = build_masks(rings, 20, True)
    B = len(rings)
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    rng = np.random.default_rng(63)
    z = (rng.standard_normal((B, N, 3 + F)).astype(np.float32)) * nm[:, :, None]
    z[:, :, :3] -= (z[:, :, :3].sum(1, keepdims=True) / nm.sum(1)[:, None, None]) * nm[:, :, None]
    eps = rng.standard_normal((B, N, 3 + F)).astype(np.float32)
    t = np.full(B, np.float32(s + 1) / np.float32(T), np.float32)
    w = np.array([0, -1, 0, 0, 0], np.float32)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    eng = _engine(eargs, esd, pargs, psd)
    got = eng.phi(z, t, nm, em)
    assert eng.kernel_variant()[1] == 4 and eng.node_buffers_global()
    assert rel_err(got, O.edm_phi(esd, eargs, z, t, nm3, em_flat)) < 1e-4
    zs = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
    assert rel_err(zs, O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm3, em_flat, eps, w, 0.6)) < 1e-4
    eng.close()


# ------------------------------------------------------------------------------------------------ V8G with P and Q in LDS
@pytest.mark.parametrize("widths", ["tiny", "default"])
def test_v8g_hybrid_residency_agrees_with_all_global(widths):
    """Round 6: a V8G launch keeps P and Q -- the two node buffers the edge phases gather from -- in LDS where that plan fits
    (kern8gp_*.hip), the other three in the global scratch as before; GAUDI_GN8_PQ=0 is round 5's all-global form.  Same source,
    two pointers in another address space: phi, predictor + gradient, a guided step and a guided chain agree to 5e-6 (the bar
    V8G holds against the resident kernels), each form is reproducible bit for bit, and the form that ran is reported."""
    from oracle import gaudi_oracle as O
    from gaudi_amd.sampling_edm import build_masks
    from tests.helpers import TINY, TINY_P
    T = 6
    F = synth.num_node_features("hetro")
    over_e, over_p = (TINY, TINY_P) if widths == "tiny" else ({}, {})
    eargs, pargs = synth.edm_args(dataset="hetro", diffusion_steps=T, **over_e), synth.pred_args(dataset="hetro", **over_p)
    esd = synth.synth_edm_state_dict(eargs, F, seed=21, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=22, amplify_coord=True)
    rings, pad = ([3, 10, 6, 8], 10) if widths == "tiny" else ([20, 13, 17, 6], 20)  # (the tiny widths' ring holds 20 nodes' split copy, not 40)
    nm3, em_flat, N = build_masks(rings, pad, True)
    B = len(rings)
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    rng = np.random.default_rng(4)
    z = O._combined_noise(rng.standard_normal((B, N, 3 + F)).astype(np.float32), nm3)
    eps = rng.standard_normal((B, N, 3 + F)).astype(np.float32)
    t = np.full(B, 0.5, np.float32)
    w = np.array([3, 0, 1, 1, 0], np.float32)
    dp = np.broadcast_to(w * np.float32(0.6), (B, 5)).copy()
    outs = []
    for env, form in (({"GAUDI_FORCE_GN8": 1}, 2), ({"GAUDI_FORCE_GN8": 1, "GAUDI_GN8_PQ": 0}, 1)):
        eng = _engine(eargs, esd, pargs, psd, **env)
        phi = eng.phi(z, t, nm, em)
        assert eng.node_buffers_form() == form and eng.kernel_variant()[1] == 8
        pred, grad = eng.predictor_grad(z, t, nm, em, dp)
        assert eng.node_buffers_form() == form
        zs = eng.step(2, z, nm, em, eps, target_w=w, scale=0.6)
        assert eng.node_buffers_form() == form
        x, h, d = eng.sample(nm, em, seed=5, target_w=w, scale=0.6)
        xb, hb, _ = eng.sample(nm, em, seed=5, target_w=w, scale=0.6)
        assert eng.node_buffers_form() == form and np.array_equal(x, xb) and np.array_equal(h, hb)
        outs.append((phi, pred, grad, zs, x))
        eng.close()
    for a, b in zip(*outs[:2]):
        assert rel_err(a, b) < 5e-6, rel_err(a, b)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    assert rel_err(outs[0][0], O.edm_phi(esd, eargs, z, t, nm3, em_flat)) < 1e-4
    assert rel_err(outs[0][3], O.step_guided(esd, eargs, psd, pargs, gamma, 2, z, nm3, em_flat, eps, w, 0.6)) < 1e-4


def test_v8g_hybrid_falls_back_to_all_global_where_it_does_not_fit():
    """A complete graph of 30 nodes at the default widths (870 edge slots: 35 KB of per-slot arrays) does not leave room for P and Q
    beside the ring: the call runs the all-global V8G form (8 waves) -- not the 4-wave kernels, as the compile-time switch of the
    first experiment did -- and the sparse 40-node hetero molecule beside it in another call runs the hybrid."""
    from oracle import gaudi_oracle as O
    eargs, pargs = synth.edm_args(diffusion_steps=1000), synth.pred_args()
    esd = synth.synth_edm_state_dict(eargs, 1, seed=41, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=42, amplify_coord=True)
    N = 30
    nm, em = O.build_masks([N, N - 7], N, False)
    rng = np.random.default_rng(N)
    z = O._combined_noise(rng.standard_normal((2, N, 4)).astype(np.float32), nm)
    eps = rng.standard_normal((2, N, 4)).astype(np.float32)
    w = O.target_max_gap_weights(5)
    gamma = O.gamma_table("polynomial_2", 1000, 1e-5)
    eng = _engine(eargs, esd, pargs, psd)
    zs = eng.step(300, z, nm, em, eps, target_w=w, scale=0.6)
    assert eng.kernel_variant()[1] == 8 and eng.node_buffers_form() == 1
    assert rel_err(zs, O.step_guided(esd, eargs, psd, pargs, gamma, 300, z, nm, em, eps, w, 0.6)) < 1e-4
    eng.close()
    eargs, esd, pargs, psd, F, nm, em, N = _hetero_batch([20, 12], 20, 1000)
    eng = _engine(eargs, esd, pargs, psd)
    rng = np.random.default_rng(1)
    z = O._combined_noise(rng.standard_normal((2, N, 3 + F)).astype(np.float32), nm[:, :, None])
    eng.step(300, z, nm, em, rng.standard_normal((2, N, 3 + F)).astype(np.float32), target_w=np.array([0, -1, 0, 0, 0], np.float32), scale=0.6)
    assert eng.kernel_variant()[1] == 8 and eng.node_buffers_form() == 2
    eng.close()


# ------------------------------------------------------------------------------------------------ attention / tanh switches (g23)
G23_NAMES = [f"{ds}_tiny_att{a}_tanh{t}" for a, t in ((0, 1), (1, 0), (0, 0)) for ds in ("cata", "hetro")] + ["cata_default_att0_tanh0"]


@pytest.mark.parametrize("family", ["8 waves, fp16 pairs", "8 waves, fp32 instructions", "4 waves", "global node buffers"])
@pytest.mark.parametrize("name", G23_NAMES)
def test_attention_and_tanh_switches_vs_reference(golden, name, family):
    """`--attention False` / `--tanh False` (utils/args_edm.py:29-30, cond_prediction/prediction_args.py:44-45) are constructor
    switches of both networks that no fixture exercised through round 5 (every golden had the defaults True / True).  g23: the
    REFERENCE's phi, predictor + input gradient and teacher-forced unguided / guided step for the three other combinations, on
    every kernel family, at 1e-4."""
    from tests.test_oracle_golden import _g23_case
    g = golden("g23_attention_tanh_flags")
    cfg, eargs, esd, pargs, psd = _g23_case(g, name)
    env = {"8 waves, fp16 pairs": {}, "8 waves, fp32 instructions": {"GAUDI_EDGE_MATH": "fp32"}, "4 waves": {"GAUDI_WAVES": 4},
           "global node buffers": {"GAUDI_FORCE_GN8": 1}}[family]
    z, t, nm, em, eps = (g[f"{name}_{k}"] for k in ("z", "t", "node_mask", "edge_mask", "step_noise"))
    B, N = z.shape[:2]
    nm2, em3 = nm.reshape(B, N), em.reshape(B, N, N)
    w = np.array([0, -1, 0, 0, 0], np.float32)
    eng = _engine(eargs, esd, pargs, psd, **env)
    assert rel_err(eng.phi(z, t[:, 0], nm2, em3), g[name + "_eps"]) < 1e-4
    assert eng.kernel_variant()[1] == (4 if family == "4 waves" else 8)
    pred, grad = eng.predictor_grad(z, t[:, 0], nm2, em3, np.broadcast_to(w * np.float32(0.6), (B, 5)).copy())
    assert rel_err(pred, g[name + "_pred"]) < 1e-4 and rel_err(grad, g[name + "_grad_gap"]) < 1e-4
    assert np.abs(grad * (1 - nm2[:, :, None])).max() == 0
    assert rel_err(eng.step(cfg["s"], z, nm2, em3, eps), g[name + "_zs_unguided"]) < 1e-4
    assert rel_err(eng.step(cfg["s"], z, nm2, em3, eps, target_w=w, scale=0.6), g[name + "_zs_guided"]) < 1e-4
    if family == "global node buffers":
        assert eng.node_buffers_global()
    eng.close()


# ------------------------------------------------------------------------------------------------ non-default scalars (g24)
@pytest.mark.parametrize("name", ["cata", "hetro"])
def test_scalar_hyperparameters_away_from_their_defaults_vs_reference(golden, name):
    """polynomial_3 with precision 1e-4, normalize_factors [2, 3, 5], coords_range 7 (denoiser) and 4 (predictor), norm_constant 2,
    normalization_factor 2, inv_sublayers 2 -- every scalar of args.txt the path reads, away from its default at once (golden g24:
    the reference's gamma table, phi, predictor + gradient, guided T = 50 chain through sample_guidance) on 8 and 4 waves."""
    from tests.test_oracle_golden import _g24_case
    g = golden("g24_scalar_hyperparameters")
    cfg, eargs, esd, pargs, psd = _g24_case(g, name, chain=False)
    c_cfg, c_eargs, c_esd, c_pargs, c_psd = _g24_case(g, name, chain=True)
    z, t, nm, em = g[name + "_z"], g[name + "_t"][:, 0], g[name + "_node_mask"], g[name + "_edge_mask"]
    B, N = z.shape[:2]
    w = np.array([0, -1, 0, 0, 0], np.float32)
    for env in ({}, {"GAUDI_WAVES": 4}):
        eng = _engine(eargs, esd, pargs, psd, **env)
        assert np.allclose(eng.gamma(), g[f"gamma_T{cfg['T']}"], rtol=3e-7, atol=0)
        assert rel_err(eng.phi(z, t, nm.reshape(B, N), em.reshape(B, N, N)), g[name + "_eps"]) < 1e-4
        pred, grad = eng.predictor_grad(z, t, nm.reshape(B, N), em.reshape(B, N, N), np.broadcast_to(w * np.float32(0.6), (B, 5)).copy())
        assert rel_err(pred, g[name + "_pred"]) < 1e-4 and rel_err(grad, g[name + "_grad_gap"]) < 1e-4
        eng.close()
        eng = _engine(c_eargs, c_esd, c_pargs, c_psd, **env)
        cnm = g[name + "_chain_node_mask"]
        Bc, Nc = cnm.shape[0], cnm.shape[1]
        x, h, d = eng.sample(cnm.reshape(Bc, Nc), g[name + "_chain_edge_mask"].reshape(Bc, Nc, Nc), noise=g[name + "_noise"], target_w=w, scale=0.6)
        assert rel_err(x, g[name + "_x_guided"]) < 1e-4 and np.array_equal(h, g[name + "_h_guided"])
        eng.close()


@pytest.mark.parametrize("nf", [64, 196])
def test_sin_embedding_other_widths_guided_as_two_launches(nf):
    """A sin_embedding denoiser has a fused (denoiser + predictor) kernel at the tiny and the default width pairs only; at every other
    hidden size of the 4-wave family it runs on its denoiser-only kernel and a guided step is two launches (denoiser, then the ordinary
    predictor-only kernel: the path large molecules take).  phi, a guided teacher-forced step and a short guided chain against the
    oracle (pinned on the reference for sin_embedding by g22) at 1e-4."""
    from oracle import gaudi_oracle as O
    from tests.helpers import TINY_P
    T, s = 8, 3
    ds = "hetro"
    F = synth.num_node_features(ds)
    eargs = synth.edm_args(dataset=ds, diffusion_steps=T, sin_embedding=True, nf=nf, n_layers=2)
    pargs = synth.pred_args(dataset=ds, **TINY_P)
    esd = synth.synth_edm_state_dict(eargs, F, seed=71)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=72)
    nm3, em_flat = O.build_masks([3, 5, 4], 5, True)
    B, N = nm3.shape[0], nm3.shape[1]
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    rng = np.random.default_rng(73)
    z = O._combined_noise(rng.standard_normal((B, N, 3 + F)).astype(np.float32), nm3)
    eps = rng.standard_normal((B, N, 3 + F)).astype(np.float32)
    t = np.full(B, np.float32(s + 1) / np.float32(T), np.float32)
    w = np.array([0, -1, 0, 0, 0], np.float32)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    eng = _engine(eargs, esd, pargs, psd)
    assert rel_err(eng.phi(z, t, nm, em), O.edm_phi(esd, eargs, z, t, nm3, em_flat)) < 1e-4
    assert eng.kernel_variant()[1] == 4 and not eng.node_buffers_global()
    assert rel_err(eng.step(s, z, nm, em, eps), O.step_unguided(esd, eargs, gamma, s, z, nm3, em_flat, eps)) < 1e-4
    zs = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
    assert rel_err(zs, O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm3, em_flat, eps, w, 0.6)) < 1e-4
    noise = rng.standard_normal((T + 2, B, N, 3 + F)).astype(np.float32)
    x, h, d = eng.sample(nm, em, noise=noise, target_w=w, scale=0.6)
    xo, ho, _ = O.sample(esd, eargs, nm3, em_flat, noise, std=1.0, pred_sd=psd, pcfg=pargs, target_w=w, scale=0.6)
    assert rel_err(x, xo) < 1e-4 and np.array_equal(h, ho)
    eng.close()


def test_mixed_wide_launch_pairs_first_then_single_molecules():
    """Batch sizes between the rounds of the chip's CUs: 640 cata molecules on 256 CUs run as ONE launch of 256 two-molecule groups and
    128 single ones in the same wide kernel (a list schedule of 2.9 round-times against 3 for one molecule per workgroup and 3.6 for
    320 pairs: gaudi_hip.hip, stage_graph8) -- and every molecule gets the bits it gets alone in its workgroup."""
    from oracle import gaudi_oracle as O
    from tests.helpers import TINY, TINY_P
    T = 3
    eargs, pargs = synth.edm_args(diffusion_steps=T, **TINY), synth.pred_args(**TINY_P)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=81)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=82)
    B = 640
    nm, em = O.build_masks([11] * B, 11, False)
    w = O.target_max_gap_weights(5)
    res = []
    for env, groups in (({"GAUDI_PAIRS": 0}, B), ({}, 384), ({"GAUDI_PAIRS": 2}, 320)):
        eng = _engine(eargs, esd, pargs, psd, **env)
        x, h, _ = eng.sample(nm, em, seed=6, target_w=w, scale=0.6)
        wg, slots = eng.last_launch_shape()
        # (384 on the 256 CUs of an MI355X; the forced forms do not depend on the CU count)
        assert wg == groups or (not env and wg in (B, 320)), (env, wg, slots)
        res.append((x, h))
        eng.close()
    for x, h in res[1:]:
        assert np.array_equal(res[0][0], x) and np.array_equal(res[0][1], h)
