"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute calls)."""
import json
import os
import types

import numpy as np
import pytest

from gaudi_amd import checkpoint, sampling_edm, synth
from gaudi_amd.models_edm import LinearTarget, PropertyNorm, target_function_max_gap, target_function_opv


class RecordingModel:
    """Captures what the sampling helpers hand to model.sample / model.sample_guidance."""

    def __init__(self):
        self.rec = {}

    def sample(self, B, n_nodes, node_mask, edge_mask, std=1.0):
        self.rec = dict(kind="sample", B=B, N=n_nodes, node_mask=node_mask, edge_mask=edge_mask, std=std)
        return np.zeros((B, n_nodes, 3), np.float32), {"categorical": np.zeros((B, n_nodes, 1), np.float32)}

    def sample_guidance(self, B, tf, node_mask, edge_mask, scale, fix_noise=False, std=1.0):
        self.rec = dict(kind="guidance", B=B, node_mask=node_mask, edge_mask=edge_mask, std=std, scale=scale, tf=tf)
        N = node_mask.shape[1]
        return np.zeros((B, N, 3), np.float32), {"categorical": np.zeros((B, N, 1), np.float32)}


@pytest.mark.parametrize("name,n_key,dataset,mx", [("cata", "cata_n", "cata", 11), ("hetro_pos", "hetro_pos_n", "hetro", 10)])
def test_sample_pos_edm_masks_match_reference(golden, name, n_key, dataset, mx):
    g = golden("g2_masks")
    args = types.SimpleNamespace(device="cpu", dataset=dataset, max_nodes=mx)
    m = RecordingModel()
    x, h, nm, em = sampling_edm.sample_pos_edm(args, m, g[n_key])
    assert m.rec["kind"] == "sample" and m.rec["std"] == 0.7  # reference default (sampling_edm.py:128)
    assert np.array_equal(np.asarray(nm), g[name + "_node_mask"])
    assert np.array_equal(np.asarray(em), g[name + "_edge_mask"])
    assert m.rec["N"] == g[name + "_node_mask"].shape[1]


@pytest.mark.parametrize("name,n_key,dataset", [("cata_guid", "cata_guid_n", "cata"), ("hetro_guid", "hetro_guid_n", "hetro")])
def test_sample_guidance_masks_match_reference(golden, name, n_key, dataset):
    g = golden("g2_masks")
    args = types.SimpleNamespace(device="cpu", dataset=dataset, max_nodes=11)
    m = RecordingModel()
    x, h, nm, em = sampling_edm.sample_guidance(args, m, "TF", g[n_key], scale=0.6)
    assert m.rec["kind"] == "guidance" and m.rec["std"] == 1.0 and m.rec["scale"] == 0.6 and m.rec["tf"] == "TF"
    assert np.array_equal(np.asarray(nm), g[name + "_node_mask"])  # padded to the BATCH max (sampling_edm.py:177)
    assert np.array_equal(np.asarray(em), g[name + "_edge_mask"])


def test_sample_pos_edm_rejects_too_many_nodes():
    args = types.SimpleNamespace(device="cpu", dataset="cata", max_nodes=5)
    with pytest.raises(AssertionError):
        sampling_edm.sample_pos_edm(args, RecordingModel(), [6])


def test_masking_assert_fires():
    class Leaky(RecordingModel):
        def sample(self, B, n_nodes, node_mask, edge_mask, std=1.0):
            return np.ones((B, n_nodes, 3), np.float32), {"categorical": np.zeros((B, n_nodes, 1), np.float32)}
    args = types.SimpleNamespace(device="cpu", dataset="cata", max_nodes=5)
    with pytest.raises(AssertionError, match="not masked"):
        sampling_edm.sample_pos_edm(args, Leaky(), [3])


@pytest.mark.parametrize("dp", [True, False])
def test_checkpoint_roundtrip(tmp_path, dp):
    args = synth.edm_args(dp=dp, nf=32, n_layers=2)
    sd = synth.synth_edm_state_dict(args, 1, seed=5)
    synth.write_checkpoint(str(tmp_path), args, sd)
    a = checkpoint.get_edm_args(str(tmp_path))
    assert a.restore is True and a.exp_dir == str(tmp_path) and a.nf == 32 and a.dp == dp
    import torch
    raw = torch.load(os.path.join(str(tmp_path), "model.pt"))
    assert all(k.startswith("module.") for k in raw) == dp
    got = checkpoint.load_state_dict(str(tmp_path))
    assert set(got) == set(sd)
    for k in sd:
        assert np.array_equal(got[k], sd[k])


def test_linear_targets():
    pred = types.SimpleNamespace(K=5, engine=None)
    t = target_function_max_gap(pred)
    assert t.weights.tolist() == [0, -1, 0, 0, 0]
    pn = PropertyNorm(mean=[0.3, -1.0, 0.5, 2.0, 0.1], std=[1.5, 0.7, 2.0, 0.9, 1.1])
    t = target_function_opv(pred, pn)
    p = np.random.default_rng(0).standard_normal((7, 5)).astype(np.float32)
    u = pn.unnormalize(p)
    want = u[:, 3] + u[:, 2] + 3 * u[:, 0]  # generation_guidance.py:205-211
    np.testing.assert_allclose(p @ t.weights + t.const, want, rtol=1e-5)


def test_unsupported_checkpoint_modes_are_refused():
    from gaudi_amd.engine import _noise_power
    from gaudi_amd._lib import GaudiError
    assert _noise_power("polynomial_2") == 2.0
    assert _noise_power("cosine") == 0.0  # PredefinedNoiseSchedule's other mode (en_diffusion.py:196-197), a host table
    for bad in ("learned", "polynomial", "polynomial_0", "linear_2", "polynomial_x", "polynomial_2_3"):  # (a bare ValueError)
        with pytest.raises(GaudiError):
            _noise_power(bad)


def test_ring_count_sampler_matches_reference(golden):
    """DistributionRings: same histogram order, probabilities, seeded Categorical draws and log-probs as the reference."""
    import torch
    from gaudi_amd.models_edm import DistributionRings
    g = golden("g12_ring_count_sampler")
    for ds in ("cata", "hetro"):
        d = DistributionRings(ds)
        assert np.array_equal(d.n_nodes.numpy(), g[f"{ds}_n_nodes"])
        assert np.array_equal(d.prob.numpy(), g[f"{ds}_prob"])
        torch.manual_seed(1234)
        s = d.sample(2000)
        assert np.array_equal(s.numpy(), g[f"{ds}_sample"])
        assert np.array_equal(d.log_prob(s[:64]).numpy(), g[f"{ds}_log_prob"])


def test_affine_target_probe():
    """GaudiModel turns a reference-form closure into the fused LinearTarget only when it is affine in the predictor outputs
    with ONE weight vector for every molecule and every t (models_edm.affine_target_weights); anything else keeps the
    callback path.  Device-free: the probe works on fn(pred, t)."""
    import torch
    from gaudi_amd.models_edm import affine_target_weights as A
    std, mean = torch.tensor([1.0, 2, 3, 4, 5]), torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
    np.testing.assert_array_equal(A(lambda p, t: -p[:, 1], 5, 1000), [0, -1, 0, 0, 0])  # generation_guidance.py:200-203
    u = lambda p: p * std + mean
    np.testing.assert_array_equal(A(lambda p, t: u(p)[:, 3] + u(p)[:, 2] + 3 * u(p)[:, 0], 5, 1000), [3, 0, 3, 4, 0])  # :205-211
    assert A(lambda p, t: 0.5 * torch.log1p(p[:, 1] ** 2), 5, 1000) is None        # not affine
    assert A(lambda p, t: t * p[:, 2], 5, 1000) is None                            # depends on t
    assert A(lambda p, t: p[:, 2] * torch.arange(3.0), 5, 1000) is None            # a different weight per molecule
    assert A(lambda p, t: torch.zeros(3), 5, 1000) is None                         # does not depend on pred
    assert A(lambda p, t: p[:, 0].abs(), 5, 1000) is None                          # piecewise linear
    assert A(lambda p, t: (p[:, 0] / 0.0), 5, 1000) is None                        # non-finite gradient
    # kinks whose backward pass is a mask (no grad_fn on the gradient) and that random probes do not reach
    assert A(lambda p, t: -torch.clamp(p[:, 1], -5e3, 5e3), 5, 1000) is None
    assert A(lambda p, t: torch.where(p[:, 1] > 7e3, 2 * p[:, 1], p[:, 1]), 5, 1000) is None
    # ... a closure that is switched off at the probed t (w = 0 would run an unguided chain without a word)
    assert A(lambda p, t: (-p[:, 1] if 0.1 < t < 0.4 else 0 * p[:, 1]), 5, 1000) is None
    # ... and one whose weight changes inside a window the three-point probe misses: recognised, then caught by the all-t check
    from gaudi_amd.models_edm import affine_gradient_holds as H
    windowed = lambda p, t: (-p[:, 1] if not (0.6 < t < 0.9) else -2 * p[:, 1])
    w = A(windowed, 5, 1000)
    np.testing.assert_array_equal(w, [0, -1, 0, 0, 0])
    assert not H(windowed, w, 1000)
    assert H(lambda p, t: -p[:, 1], w, 1000)


def test_property_norm_accepts_torch_predictions_that_require_grad():
    """The reference's OPV closure calls prop_dist.unnormalize(pred) on the predictor's torch output inside autograd
    (generation_guidance.py:205-211, models_edm.py:186-192): the mirror's PropertyNorm must stay differentiable there -- and the
    closure must come out as the affine target it is."""
    import torch
    from gaudi_amd.models_edm import PropertyNorm, affine_target_weights
    pd = PropertyNorm(np.array([0.1, -0.2, 0.3, 0.4, 0.5]), np.array([1.5, 0.5, 2.0, 0.7, 1.0]))
    p = torch.randn(3, 5, requires_grad=True)
    u = pd.unnormalize(p)
    assert torch.is_tensor(u) and u.requires_grad and torch.allclose(pd.normalize(u), p, atol=1e-6)
    assert isinstance(pd.unnormalize(np.ones((2, 5), np.float32)), np.ndarray)

    def opv(pred, t):  # generation_guidance.py:205-211
        pred = pd.unnormalize(pred)
        gap, ea, ip = pred[:, 0], pred[:, 2], pred[:, 3]
        return ip + ea + 3 * gap

    np.testing.assert_allclose(affine_target_weights(opv, 5, 1000), [4.5, 0, 2.0, 0.7, 0], rtol=1e-6)


def test_mfma_count_model_three_column_tile_pass():
    """The V8G kernels run 33..48 node columns as ONE pass over three 16-column tiles (w8_common.h: node_gemm_n,
    w8_nodes_f16.h: node_gemm_h); the resident kernels' model (pairs of tiles) is unchanged.  Edge-level counts do not depend on
    it."""
    from gaudi_amd import flops, synth
    e, p = synth.edm_args(dataset="hetro"), synth.pred_args(dataset="hetro")
    for variant in ("w8", "w8s"):  # fp32 node GEMMs / fp16-pair node GEMMs (16-bit instructions + fp32 tail steps)
        base = flops.step_mfma_counts(14, 40, e, p, variant)
        v8g = flops.step_mfma_counts(14, 40, e, p, variant, 3)
        one = flops.step_mfma_counts(14, 16, e, p, variant, 3)
        two = flops.step_mfma_counts(14, 30, e, p, variant, 3)
        # the node part of both instruction classes scales with the column tiles, 4 : 3 : 2 : 1; the edge part does not depend on them
        for k in (0, 1):
            edge = one[k] - (two[k] - one[k])
            if one[k] == edge:
                assert base[k] == v8g[k] == one[k]
                continue
            assert abs((base[k] - edge) / (one[k] - edge) - 4) < 1e-9 and abs((v8g[k] - edge) / (one[k] - edge) - 3) < 1e-9
    assert flops.step_mfma_counts(14, 50, e, p, "w8s", 3) == flops.step_mfma_counts(14, 50, e, p, "w8s")  # 4 tiles either way
