"""Pin the C++/OpenMP restatement (oracle/gaudi_cpu.cpp: the CPU baseline bench.py reports a GPU number beside, and a second
checker next to the numpy oracle) against the golden vectors the REFERENCE produced (tools/make_golden.py): phi (g3),
predictor + input gradient (g4), teacher-forced unguided / guided steps incl. the clip branch (g5).  CPU-only."""
import numpy as np
import pytest

from oracle import build_cpu
from oracle import gaudi_oracle as O
from tests.helpers import TINY, TINY_P, cfg_of, edm_from_cfg, max_norm_err, pred_from_cfg, rel_err

pytestmark = pytest.mark.skipif(not build_cpu.cpu_ok(), reason="the CPU port is built for AVX2 + FMA hosts")


@pytest.fixture(scope="module")
def port():
    p = build_cpu.CpuPort()
    yield p
    p.close()


@pytest.mark.parametrize("name", ["cata_tiny", "cata_tiny_amp", "hetro_tiny_amp", "cata_tiny_sub2_amp", "cata_full", "hetro_full_amp"])
def test_phi_vs_reference(golden, port, name):
    g = golden("g3_phi")
    args, sd = edm_from_cfg(cfg_of(g, name))
    port.load_edm(args, sd)
    eps = port.phi(g[name + "_z"], g[name + "_t"][:, 0], g[name + "_node_mask"], g[name + "_edge_mask"])
    assert max_norm_err(eps, g[name + "_eps"]) < 1e-5 and rel_err(eps, g[name + "_eps"]) < 1e-4
    assert np.abs(eps * (1 - g[name + "_node_mask"])).max() == 0  # masked nodes output exactly zero


@pytest.mark.parametrize("name", ["cata_tiny_amp", "hetro_tiny_amp", "cata_full", "hetro_full_amp"])
def test_predictor_and_gradient_vs_reference(golden, port, name):
    g = golden("g4_predictor")
    args, sd = pred_from_cfg(cfg_of(g, name))
    port.load_predictor(args, sd)
    z, t, nm, em = g[name + "_z"], g[name + "_t"][:, 0], g[name + "_node_mask"], g[name + "_edge_mask"]
    assert rel_err(port.predictor(z, t, nm, em), g[name + "_pred"]) < 1e-5
    for tn, w in (("gap", O.target_max_gap_weights(5)), ("opv", O.target_opv_weights(5, g["prop_std"]))):
        pred, grad = port.predictor(z, t, nm, em, dpred=w * g["scale"])
        assert rel_err(pred, g[name + "_pred"]) < 1e-5
        assert rel_err(grad, g[f"{name}_grad_{tn}"]) < 2e-5, tn
        assert np.abs(grad * (1 - nm)).max() == 0  # masked nodes get exactly zero gradient


@pytest.mark.parametrize("name", ["cata", "hetro"])
def test_teacher_forced_steps_vs_reference(golden, port, name):
    g = golden("g5_steps")
    cfg = cfg_of(g, name)
    T = cfg["T"]
    eargs, esd = edm_from_cfg(dict(dataset=cfg["dataset"], over=TINY, wseed=cfg["eseed"], amp=True), diffusion_steps=T)
    pargs, psd = pred_from_cfg(dict(dataset=cfg["dataset"], over=TINY_P, wseed=cfg["pseed"], amp=True))
    port.load_edm(eargs, esd)
    port.load_predictor(pargs, psd)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    z, nm, em = g[name + "_z"], g[name + "_node_mask"], g[name + "_edge_mask"]
    w = O.target_max_gap_weights(5)
    for s in (0, 1, 500, 998, 999):
        eps = g[f"{name}_s{s}_eps"]
        c = O.step_coefficients(gamma, s, s + 1)
        t_val = np.float32(np.float32(s + 1) / np.float32(T))
        assert rel_err(port.step(c, t_val, z, nm, em, eps), g[f"{name}_s{s}_zs_unguided"]) < 1e-5, s
        for scale in (0.6, 400.0):
            zg = port.step(c, t_val, z, nm, em, eps, target_w=w, scale=scale)
            assert rel_err(zg, g[f"{name}_s{s}_zs_guided_scale{scale}"]) < 2e-5, (s, scale)


def test_port_agrees_with_numpy_oracle_on_a_default_size_guided_step(port):
    """Default architectures (nf 192 / 196, 9 / 12 layers), B = 4 mixed sizes: the two CPU restatements agree."""
    from gaudi_amd import synth
    T = 1000
    eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
    esd, psd = synth.synth_edm_state_dict(eargs, 1, seed=0), synth.synth_predictor_state_dict(pargs, 1, 5, seed=1)
    port.load_edm(eargs, esd)
    port.load_predictor(pargs, psd)
    nm, em = O.build_masks([11, 7, 11, 4], 11, False)
    rng = np.random.default_rng(5)
    z = O._combined_noise(rng.standard_normal((4, 11, 4)).astype(np.float32), nm)
    eps = rng.standard_normal((4, 11, 4)).astype(np.float32)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    w = O.target_max_gap_weights(5)
    s = 700
    want = O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6)
    got = port.step(O.step_coefficients(gamma, s, s + 1), np.float32(np.float32(s + 1) / np.float32(T)), z, nm, em, eps, target_w=w, scale=0.6)
    assert rel_err(got, want) < 2e-5


def test_molecule_groups_do_not_change_a_molecule():
    """The port runs groups of molecules layer by layer (one GEMM per Linear over the group's rows, so that a weight matrix is
    read once per group: the CPU baseline scales with the cores instead of with the memory system).  A molecule's result must
    not depend on the group it is in: groups of 1 (the round-3 arrangement), 3 and 8 agree bit for bit on a ragged hetero batch,
    guided step and predictor gradient."""
    from oracle import build_cpu
    if not build_cpu.cpu_ok():
        pytest.skip("host CPU lacks AVX2/FMA")
    from gaudi_amd import synth
    from gaudi_amd.sampling_edm import build_masks
    T = 100
    F = synth.num_node_features("hetro")
    eargs, pargs = synth.edm_args(dataset="hetro", diffusion_steps=T, nf=32, n_layers=2), synth.pred_args(dataset="hetro", nf=36, n_layers=3)
    esd = synth.synth_edm_state_dict(eargs, F, seed=3, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=4, amplify_coord=True)
    rings = [3, 10, 4, 7, 5, 9, 3, 6, 8, 10, 4]
    nm3, em_flat, N = build_masks(rings, 10, True)
    B = len(rings)
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    rng = np.random.default_rng(1)
    z = O._combined_noise(rng.standard_normal((B, N, 3 + F)).astype(np.float32), nm3)
    eps = rng.standard_normal((B, N, 3 + F)).astype(np.float32)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    w = np.array([3, 0, 1, 1, 0], np.float32)
    port = build_cpu.CpuPort()
    port.load_edm(eargs, esd)
    port.load_predictor(pargs, psd)
    outs = []
    for g in (1, 3, 8):
        port.set_group(g)
        port.set_threads(2)  # few threads -> the groups really hold several molecules
        zs = port.step(O.step_coefficients(gamma, 40, 41), np.float32(0.41), z, nm, em, eps, target_w=w, scale=0.6)
        pred, grad = port.predictor(z, 0.41, nm, em, dpred=w)
        outs.append((zs, pred, grad, port.phi(z, 0.41, nm, em)))
    port.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)
    want = O.step_guided(esd, eargs, psd, pargs, gamma, 40, z, nm3, em, eps, w, 0.6)
    assert rel_err(outs[0][0], want) < 1e-4


@pytest.mark.parametrize("name", ["cata_tiny_att0_tanh1", "hetro_tiny_att1_tanh0", "hetro_tiny_att0_tanh0", "cata_default_att0_tanh0"])
def test_attention_and_tanh_switches_vs_reference(golden, port, name):
    """`--attention False` / `--tanh False` on both networks (g23): the C++ port carries the switches too."""
    from tests.test_oracle_golden import _g23_case
    g = golden("g23_attention_tanh_flags")
    cfg, eargs, esd, pargs, psd = _g23_case(g, name)
    port.load_edm(eargs, esd)
    port.load_predictor(pargs, psd)
    z, t, nm, em, eps = (g[f"{name}_{k}"] for k in ("z", "t", "node_mask", "edge_mask", "step_noise"))
    w = np.array([0, -1, 0, 0, 0], np.float32)
    assert rel_err(port.phi(z, t[:, 0], nm, em), g[name + "_eps"]) < 1e-5
    pred, grad = port.predictor(z, t[:, 0], nm, em, dpred=w * np.float32(0.6))
    assert rel_err(pred, g[name + "_pred"]) < 1e-5 and rel_err(grad, g[name + "_grad_gap"]) < 2e-5
    gamma = O.gamma_table("polynomial_2", cfg["T"], 1e-5)
    s = cfg["s"]
    c = O.step_coefficients(gamma, s, s + 1)
    t_val = np.float32(np.float32(s + 1) / np.float32(cfg["T"]))
    assert rel_err(port.step(c, t_val, z, nm, em, eps), g[name + "_zs_unguided"]) < 1e-5
    assert rel_err(port.step(c, t_val, z, nm, em, eps, target_w=w, scale=0.6), g[name + "_zs_guided"]) < 2e-5
