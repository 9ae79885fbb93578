"""GEMMs on the 16-bit matrix pipe with exactly split fp32 operands (gaudi_amd/csrc/w8_split.h, w8_nodes_f16.h: pairs of fp16
pieces since round 5, three bf16 pieces in rounds 2-4), the default of the 8-wave kernels.  The whole GPU suite runs on it at unchanged tolerances; here: the switch and its fallback, the error of the
split form against a float64 evaluation next to the error of the fp32 matrix instruction, and the core parity tests repeated
on GAUDI_EDGE_MATH=fp32 so the fp32-instruction kernels stay covered."""
import numpy as np
import pytest

from gaudi_amd import synth
from tests import test_gpu_parity as P
from tests.helpers import max_norm_err, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import gaudi_oracle
    return gaudi_oracle


@pytest.fixture
def fp32_math(monkeypatch):
    monkeypatch.delenv("GAUDI_WAVES", raising=False)
    monkeypatch.setenv("GAUDI_EDGE_MATH", "fp32")


def _engine(monkeypatch, math, eargs, esd, pargs=None, psd=None):
    from gaudi_amd.engine import Engine
    monkeypatch.delenv("GAUDI_WAVES", raising=False)
    if math is None:
        monkeypatch.delenv("GAUDI_EDGE_MATH", raising=False)
    else:
        monkeypatch.setenv("GAUDI_EDGE_MATH", math)
    eng = Engine(0)
    eng.load_edm(eargs, esd)
    if pargs is not None:
        eng.load_predictor(pargs, psd)
    return eng


def test_split_is_the_default_and_env_selects_fp32(monkeypatch, O):
    eargs = synth.edm_args(nf=64, n_layers=2, diffusion_steps=10)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=3)
    nm, em = O.build_masks([5, 7], 7, False)
    z = np.random.default_rng(0).standard_normal((2, 7, 4)).astype(np.float32) * nm
    t = np.array([0.3, 0.6], np.float32)
    want = O.edm_phi(esd, eargs, z, t, nm, em)
    for math, flag in ((None, 1), ("fp32", 0), ("split", 1)):
        eng = _engine(monkeypatch, math, eargs, esd)
        got = eng.phi(z, t, nm, em)
        assert eng.edge_math() == (flag, flag) and eng.kernel_variant() == (8, 8)
        assert rel_err(got, want) < 2e-5
        eng.close()
    # the 4-wave family has no split form
    monkeypatch.setenv("GAUDI_WAVES", "4")
    monkeypatch.delenv("GAUDI_EDGE_MATH", raising=False)
    from gaudi_amd.engine import Engine
    eng = Engine(0)
    eng.load_edm(eargs, esd)
    eng.phi(z, t, nm, em)
    assert eng.edge_math() == (0, 0)
    eng.close()


@pytest.mark.parametrize("dataset,nodes,nf_e,nf_p", [("cata", [11, 11, 7, 4, 9, 11], 192, 196), ("cata", [9, 5, 8], 36, 36),
                                                     ("hetro", [5, 3, 4, 2], 64, 60)])
def test_split_error_vs_float64_is_not_larger_than_the_fp32_instructions(monkeypatch, O, dataset, nodes, nf_e, nf_p):
    """Denoiser output, predictor output and predictor input-gradient of both arithmetic forms against the float64 evaluation
    of the oracle: the split form (two fp16 pieces per operand, three piece products, fp32 accumulate) must sit at the same
    fp32 rounding level as the fp32 matrix instruction.  Whole-network errors are single draws of rounding noise amplified
    by 9-12 layers (the 4-wave and 8-wave fp32 families differ from each other by up to 1.4x on the same inputs), hence
    the factor 3 (+ a floor of 2e-7 of the tensor's max) and the absolute bar of 1e-5.  For ONE GEMM the split form is the
    closer of the two (tools/split_gemm_microbench.hip, output in profiles/)."""
    F = synth.num_node_features(dataset)
    eargs = synth.edm_args(nf=nf_e, n_layers=3 if nf_e < 192 else 9, diffusion_steps=100, dataset=dataset)
    pargs = synth.pred_args(nf=nf_p, n_layers=3 if nf_p < 196 else 12, dataset=dataset)
    esd = synth.synth_edm_state_dict(eargs, F, seed=11, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=12, amplify_coord=True)
    nm, em = O.build_masks(nodes, max(nodes), dataset != "cata")
    B, N = nm.shape[0], nm.shape[1]
    rng = np.random.default_rng(5)
    z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm
    z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1, keepdims=True), 1) * nm
    t = np.linspace(0.15, 0.85, B).astype(np.float32)
    w = np.broadcast_to(np.array([0.5, -1.0, 0.25, 0.0, 1.0], np.float32), (B, 5))
    ref_eps = O.edm_phi(esd, eargs, z, t, nm, em, dtype=np.float64)
    ref_pred, ref_grad = O.predictor_grad(psd, pargs, z, nm, em, t, w, dtype=np.float64)
    errs = {}
    for math in ("split", "fp32"):
        eng = _engine(monkeypatch, math, eargs, esd, pargs, psd)
        eps = eng.phi(z, t, nm, em)
        pred, grad = eng.predictor_grad(z, t, nm, em, w)
        assert eng.edge_math()[1] == (1 if math == "split" else 0)
        errs[math] = (max_norm_err(eps, ref_eps), max_norm_err(pred, ref_pred), max_norm_err(grad, ref_grad))
        eng.close()
    for es, ef in zip(errs["split"], errs["fp32"]):
        assert es <= 3.0 * ef + 2e-7, errs
        assert es < 1e-5, errs


def test_split_ring_size_follows_the_lds_budget(monkeypatch, O):
    """The full split weight ring is 2x the fp32 one (two slots of a 32-input chunk in two fp16 pieces; 3x with round 2's bf16
    pieces).  A molecule whose node buffers leave no room for it runs on the half-ring form (two trips per chunk) of the same
    handle; both forms match the oracle."""
    eargs = synth.edm_args(dataset="hetro", diffusion_steps=20, n_layers=2)
    pargs = synth.pred_args(dataset="hetro", n_layers=2)
    F = synth.num_node_features("hetro")
    esd = synth.synth_edm_state_dict(eargs, F, seed=1, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=2, amplify_coord=True)
    eng = _engine(monkeypatch, None, eargs, esd, pargs, psd)
    gamma = O.gamma_table("polynomial_2", 20, 1e-5)
    w = np.array([3.0, 0.0, 1.0, 1.0, 0.0], np.float32)
    # (round 5: the fp16-pair ring is two thirds of the bf16 one -- 52 KiB instead of 78 -- and now fits beside 20 nodes' buffers;
    # whichever form the LDS plan picks for the large molecules, it must be a split form, and both match the oracle)
    for rings, want_split in (([10, 6, 9], (1, 2)), ([5, 3, 4], (1,))):
        nm, em = O.build_masks(rings, max(rings), True)
        B, N = nm.shape[0], nm.shape[1]
        rng = np.random.default_rng(3)
        z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm
        z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / nm.sum(1, keepdims=True) * nm
        eps = rng.standard_normal(z.shape).astype(np.float32)
        got = eng.step(7, z, nm, em, eps, target_w=w, scale=0.5)
        assert eng.kernel_variant() == (8, 8) and eng.edge_math()[0] == 1 and eng.edge_math()[1] in want_split, (rings, eng.edge_math())
        assert rel_err(got, O.step_guided(esd, eargs, psd, pargs, gamma, 7, z, nm, em, eps, w, 0.5)) < 1e-4
    eng.close()


# ---- the fp32-instruction 8-wave kernels under the core parity tests
@pytest.mark.parametrize("name", ["cata_tiny_amp", "hetro_tiny_amp", "cata_full", "hetro_full_amp"])
def test_fp32_math_phi(fp32_math, golden, O, name):
    P.test_phi_vs_reference(golden, O, name)


@pytest.mark.parametrize("name", ["cata_tiny_amp", "hetro_full_amp"])
def test_fp32_math_predictor(fp32_math, golden, O, name):
    P.test_predictor_forward_and_gradient(golden, O, name)


@pytest.mark.parametrize("name", ["cata", "hetro"])
def test_fp32_math_guided_steps(fp32_math, golden, O, name):
    P.test_guided_steps_teacher_forced(golden, O, name)


def test_fp32_math_chains(fp32_math, golden):
    P.test_tiny_chains_guided(golden, "hetro_tiny", 1e-4)
    P.test_c1_end_to_end_unguided(golden)
    P.test_sample_chain(golden, "cata")


def test_random_graphs_all_kernel_variants_agree():
    """tools/fuzz_split_vs_fp32.py: random widths, batch sizes, ring counts, knocked-out edges -- split (full and half ring),
    fp32-instruction and 4-wave kernels must agree to fp32 rounding level on denoiser, both reverse steps and the
    predictor gradient.  500 cases per run, failing bar 1e-5 = 3x the worst difference ever seen (3.3e-6 over 1500 cases)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_split_vs_fp32", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_split_vs_fp32.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(500, tol=1e-5) == 0
