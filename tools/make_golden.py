#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (read-only, /root/reference) on CPU.

Runs only in the build container (the GPU box has no /root/reference).  Nothing from the
reference is copied: it is imported, fed seeded synthetic weights (gaudi_amd.synth) and
injected noise, and only numeric inputs/outputs are stored.  Recipe = SURVEY.md Appendix A.

    python tools/make_golden.py            # rewrites every fixture
"""
import copy
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for m in ["rdkit", "rdkit.Chem", "rdkit.Chem.Draw", "rdkit.Chem.rdmolops", "rdkit.Chem.rdchem",
          "rdkit.Chem.AllChem", "imageio", "torch.utils.tensorboard"]:
    sys.modules[m] = MagicMock()

import torch  # noqa: E402

import models_edm  # noqa: E402  (reference)
import sampling_edm as ref_sampling  # noqa: E402  (reference)
from cond_prediction.train_cond_predictor import get_cond_predictor_model  # noqa: E402
from cond_prediction.prediction_args import PredictionArgs  # noqa: E402
from utils.args_edm import Args_EDM  # noqa: E402
from utils.helpers import switch_grad_off  # noqa: E402

from gaudi_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
META = dict(torch=torch.__version__, numpy=np.__version__, threads=torch.get_num_threads())


class FakeDataset:
    def __init__(self, F, K):
        self.num_node_features = F
        self.num_targets = K
        self.mean = torch.zeros(K)
        self.std = torch.ones(K)


class FakeLoader:
    def __init__(self, F, K):
        self.dataset = FakeDataset(F, K)


class InjectNoise:
    """Replace torch.randn by a queue of pre-drawn tensors (call order: x-noise then h-noise)."""

    def __init__(self, eps_list, dtype=None):
        # eps_list: list of [B,N,3+F] arrays, one per sample_combined_position_feature_noise call
        self.q = []
        for e in eps_list:
            e = torch.from_numpy(np.ascontiguousarray(e))
            if dtype is not None:
                e = e.to(dtype)
            self.q.append(e[:, :, :3].contiguous())
            self.q.append(e[:, :, 3:].contiguous())
        self.orig = None

    def __enter__(self):
        self.orig = torch.randn

        def fake(size, device=None, **kw):
            t = self.q.pop(0)
            assert tuple(t.shape) == tuple(size), (t.shape, size)
            return t.clone()

        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self.orig
        assert not self.q, f"{len(self.q)} injected tensors unused"


def build_ref_edm(dataset, sd_np, **over):
    a = Args_EDM().parse_args([])
    a.device = torch.device("cpu")
    a.dp = False
    a.restore = None
    a.dataset = dataset
    for k, v in over.items():
        setattr(a, k, v)
    F = synth.num_node_features(dataset)
    model, _, prop = models_edm.get_model(a, FakeLoader(F, 5), only_norm=True)
    full = model.state_dict()
    for k, v in sd_np.items():
        assert k in full and tuple(full[k].shape) == v.shape, k
        full[k] = torch.from_numpy(v.copy())
    model.load_state_dict(full)
    switch_grad_off([model])
    return a, model


def build_ref_pred(dataset, sd_np, K=5, **over):
    a = PredictionArgs().parse_args([])
    a.device = torch.device("cpu")
    a.dp = False
    a.restore = None
    a.dataset = dataset
    for k, v in over.items():
        setattr(a, k, v)
    F = synth.num_node_features(dataset)
    pred = get_cond_predictor_model(a, FakeDataset(F, K))
    full = pred.state_dict()
    assert set(full.keys()) == set(sd_np.keys()), set(full.keys()) ^ set(sd_np.keys())
    pred.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    switch_grad_off([pred])
    return a, pred


def masks(dataset, nodesxsample, max_nodes):
    """Reference mask construction, run through the reference's own code by calling
    sample_guidance / sample_pos_edm with a recording fake model."""
    rec = {}

    class Rec:
        def sample(self, B, n_nodes, node_mask, edge_mask, std=1.0):
            rec.update(node_mask=node_mask.numpy().copy(), edge_mask=edge_mask.numpy().copy())
            x = torch.zeros(B, n_nodes, 3)
            return x, {"categorical": torch.zeros(B, n_nodes, 1)}

        def sample_guidance(self, B, tf, node_mask, edge_mask, scale, fix_noise=False, std=1.0):
            rec.update(node_mask=node_mask.numpy().copy(), edge_mask=edge_mask.numpy().copy())
            x = torch.zeros(B, node_mask.shape[1], 3)
            return x, {"categorical": torch.zeros(B, node_mask.shape[1], 1)}

    class A:
        pass

    a = A()
    a.device = torch.device("cpu")
    a.dataset = dataset
    a.max_nodes = max_nodes
    n = torch.tensor(nodesxsample).long()
    if max_nodes is None:
        ref_sampling.sample_guidance(a, Rec(), None, n)
    else:
        ref_sampling.sample_pos_edm(a, Rec(), n)
    return rec["node_mask"], rec["edge_mask"]


def rng_noise(seed, shape):
    return np.random.Generator(np.random.Philox(key=seed)).standard_normal(shape).astype(np.float32)


def save(name, **arrs):
    arrs["meta"] = np.array(json.dumps(META))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------------------
def g1_schedule():
    out = {}
    for T in (50, 1000):
        sd = synth.synth_edm_state_dict(synth.edm_args(nf=8, n_layers=1), 1, seed=0)
        a, model = build_ref_edm("cata", sd, nf=8, n_layers=1, diffusion_steps=T)
        gamma = model.gamma.gamma.numpy().copy()
        out[f"gamma_T{T}"] = gamma
        rows = []
        for s in ([0, 1, T // 2, T - 2, T - 1]):
            st = torch.full((1, 1), s) / T
            tt = (torch.full((1, 1), s) + 1) / T
            gs, gt = model.gamma(st), model.gamma(tt)
            zt = torch.zeros(1, 1, 1)
            s2, s_ts, a_ts = model.sigma_and_alpha_t_given_s(gt, gs, zt)
            sig_s, sig_t = model.sigma(gs, zt), model.sigma(gt, zt)
            rows.append([s, a_ts.item(), s2.item(), (s2 / a_ts / sig_t).item(), (s_ts * sig_s / sig_t).item(),
                         sig_s.item(), sig_t.item(), tt.item()])
        out[f"coef_T{T}"] = np.array(rows, dtype=np.float64)
    save("g1_schedule", **out)


def g2_masks():
    out = {}
    nm, em = masks("cata", [4, 11, 7, 1], 11)
    out["cata_n"] = np.array([4, 11, 7, 1])
    out["cata_node_mask"], out["cata_edge_mask"] = nm, em
    nm, em = masks("hetro", [3, 5, 10, 7], 10)
    out["hetro_pos_n"] = np.array([3, 5, 10, 7])
    out["hetro_pos_node_mask"], out["hetro_pos_edge_mask"] = nm, em
    nm, em = masks("hetro", [3, 5, 9, 7], None)  # sample_guidance pads to batch max (9 -> N=18)
    out["hetro_guid_n"] = np.array([3, 5, 9, 7])
    out["hetro_guid_node_mask"], out["hetro_guid_edge_mask"] = nm, em
    nm, em = masks("cata", [6, 8, 8], None)
    out["cata_guid_n"] = np.array([6, 8, 8])
    out["cata_guid_node_mask"], out["cata_guid_edge_mask"] = nm, em
    save("g2_masks", **out)


TINY = dict(nf=32, n_layers=2)
TINY_P = dict(nf=36, n_layers=3)  # 36: not a multiple of 16 -> exercises feature padding


def case_inputs(dataset, nodes, max_nodes, seed, guidance_pad=False):
    nm, em = masks(dataset, nodes, None if guidance_pad else max_nodes)
    B, N, _ = nm.shape
    F = synth.num_node_features(dataset)
    z = rng_noise(seed, (B, N, 3 + F)) * nm
    z[:, :, :3] -= (z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1, keepdims=True), 1)) * nm
    return nm, em, z.astype(np.float32)


def g3_phi():
    out = {}
    cases = [
        ("cata_tiny", "cata", [4, 11, 7, 1, 11], 11, TINY, False),
        ("cata_tiny_amp", "cata", [4, 11, 7, 2, 11], 11, TINY, True),
        ("hetro_tiny_amp", "hetro", [3, 5, 10, 7], 10, TINY, True),
        ("cata_tiny_sub2_amp", "cata", [5, 9, 11], 11, dict(nf=32, n_layers=2, inv_sublayers=2,
                                                            normalization_factor=3.0, norm_constant=0.5), True),
        ("cata_full", "cata", [11, 8, 11], 11, dict(nf=192, n_layers=9), False),
        ("hetro_full_amp", "hetro", [10, 6], 10, dict(nf=192, n_layers=9), True),
    ]
    for i, (name, ds, nodes, mx, over, amp) in enumerate(cases):
        F = synth.num_node_features(ds)
        args = synth.edm_args(dataset=ds, **over)
        sd = synth.synth_edm_state_dict(args, F, seed=100 + i, amplify_coord=amp)
        a, model = build_ref_edm(ds, sd, **over)
        nm, em, z = case_inputs(ds, nodes, mx, seed=200 + i)
        B = z.shape[0]
        t = np.linspace(0.05, 0.95, B).astype(np.float32).reshape(B, 1)
        with torch.no_grad():
            eps = model.phi(torch.from_numpy(z), torch.from_numpy(t), torch.from_numpy(nm),
                            torch.from_numpy(em), None).numpy()
        out[name + "_cfg"] = np.array(json.dumps(dict(dataset=ds, over=over, amp=amp, wseed=100 + i)))
        out[name + "_z"], out[name + "_t"] = z, t
        out[name + "_node_mask"], out[name + "_edge_mask"] = nm, em
        out[name + "_eps"] = eps
    save("g3_phi", **out)


TARGETS = {
    "gap": lambda pred, prop: -pred[:, 1],
    "opv": lambda pred, prop: (lambda u: u[:, 3] + u[:, 2] + 3 * u[:, 0])(pred * prop["std"] + prop["mean"]),
}


def g4_predictor():
    out = {}
    cases = [
        ("cata_tiny_amp", "cata", [4, 11, 7, 1, 11], 11, TINY_P, True, False),
        ("hetro_tiny_amp", "hetro", [3, 5, 9, 7], None, TINY_P, True, True),
        ("cata_full", "cata", [11, 9], 11, dict(nf=196, n_layers=12), False, False),
        ("hetro_full_amp", "hetro", [10, 4], 10, dict(nf=196, n_layers=12), True, False),
    ]
    prop = dict(mean=torch.tensor([0.3, -1.0, 0.5, 2.0, 0.1]), std=torch.tensor([1.5, 0.7, 2.0, 0.9, 1.1]))
    for i, (name, ds, nodes, mx, over, amp, gpad) in enumerate(cases):
        F = synth.num_node_features(ds)
        pargs = synth.pred_args(dataset=ds, **over)
        sd = synth.synth_predictor_state_dict(pargs, F, 5, seed=300 + i, amplify_coord=amp)
        a, pred = build_ref_pred(ds, sd, **over)
        nm, em, z = case_inputs(ds, nodes, mx, seed=400 + i, guidance_pad=gpad)
        B = z.shape[0]
        t = np.full((B, 1), 0.37, np.float32)
        zt = torch.from_numpy(z).requires_grad_()
        p = pred(zt, torch.from_numpy(nm), torch.from_numpy(em), torch.from_numpy(t))
        out[name + "_pred"] = p.detach().numpy()
        for tn, tf in TARGETS.items():
            energy = (0.6 * tf(p, prop)).sum()
            g = torch.autograd.grad(energy, zt, retain_graph=True)[0]
            out[f"{name}_grad_{tn}"] = g.numpy()
        out[name + "_cfg"] = np.array(json.dumps(dict(dataset=ds, over=over, amp=amp, wseed=300 + i)))
        out[name + "_z"], out[name + "_t"] = z, t
        out[name + "_node_mask"], out[name + "_edge_mask"] = nm, em
    out["prop_mean"], out["prop_std"] = prop["mean"].numpy(), prop["std"].numpy()
    out["scale"] = np.float32(0.6)
    save("g4_predictor", **out)


def g5_steps():
    """Teacher-forced reverse steps (unguided + guided) at several t, tiny configs."""
    out = {}
    T = 1000
    for ci, (name, ds, nodes) in enumerate([("cata", "cata", [4, 11, 7, 11]), ("hetro", "hetro", [3, 9, 6])]):
        F = synth.num_node_features(ds)
        eargs = synth.edm_args(dataset=ds, **TINY)
        esd = synth.synth_edm_state_dict(eargs, F, seed=500 + ci, amplify_coord=True)
        a, model = build_ref_edm(ds, esd, **TINY)
        pargs = synth.pred_args(dataset=ds, **TINY_P)
        psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=510 + ci, amplify_coord=True)
        pa, pred = build_ref_pred(ds, psd, **TINY_P)
        nm, em, z = case_inputs(ds, nodes, None, seed=520 + ci, guidance_pad=True)
        B, N, D = z.shape
        tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)

        def tf_gap(_in, _nm, _em, _t):
            return -pred(_in, _nm, _em, _t)[:, 1]

        out[f"{name}_z"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = z, nm, em
        for s in (0, 1, 500, 998, 999):
            eps = rng_noise(600 + 10 * ci + s % 7, (B, N, D))
            st = torch.full((B, 1), s) / T
            tt = (torch.full((B, 1), s) + 1) / T
            with InjectNoise([eps]), torch.no_grad():
                zs = model.sample_p_zs_given_zt(st, tt, torch.from_numpy(z), tnm, tem, None).numpy()
            out[f"{name}_s{s}_eps"] = eps
            out[f"{name}_s{s}_zs_unguided"] = zs
            for scale in (0.6, 400.0):  # 400: forces the ||g||>10 clip branch
                with InjectNoise([eps]), torch.no_grad():
                    zg = model.sample_p_zs_given_zt_guidance(st, tt, torch.from_numpy(z), tnm, tem, tf_gap,
                                                             scale).numpy()
                out[f"{name}_s{s}_zs_guided_scale{scale}"] = zg
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=500 + ci, pseed=510 + ci, T=T)))
    save("g5_steps", **out)


def g6_decode():
    out = {}
    for ci, (name, ds, nodes) in enumerate([("cata", "cata", [4, 11, 7]), ("hetro", "hetro", [3, 9, 6])]):
        F = synth.num_node_features(ds)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **TINY), F, seed=700 + ci, amplify_coord=True)
        a, model = build_ref_edm(ds, esd, **TINY)
        nm, em, z = case_inputs(ds, nodes, None, seed=720 + ci, guidance_pad=True)
        eps = rng_noise(730 + ci, z.shape)
        with InjectNoise([eps]), torch.no_grad():
            x, h = model.sample_p_xh_given_z0(torch.from_numpy(z), torch.from_numpy(nm), torch.from_numpy(em), None)
        out[f"{name}_z"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"], out[f"{name}_eps"] = z, nm, em, eps
        out[f"{name}_x"], out[f"{name}_h"] = x.numpy(), h["categorical"].numpy().astype(np.float32)
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=700 + ci)))
    save("g6_decode", **out)


def g7_end_to_end():
    """C1: cata, 4 rings padded to 11, B=8, T=50, unguided, default architecture and default
    init (+ tiny guided T=50 variants).  Entry points: the reference's sample_pos_edm /
    sample_guidance with torch.randn injected."""
    out = {}
    # --- C1 (full default architecture, default init, std=1.0)
    T = 50
    over = dict(diffusion_steps=T)
    eargs = synth.edm_args(**over)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=0)
    a, model = build_ref_edm("cata", esd, **over)
    a.max_nodes = 11
    B, N, D = 8, 11, 4
    noise = rng_noise(1, (T + 2, B, N, D))
    with InjectNoise(list(noise)):
        x, h, nm, em = ref_sampling.sample_pos_edm(a, model, torch.tensor([4] * B), std=1.0)
    out["c1_noise"], out["c1_x"], out["c1_h"] = noise, x.numpy(), h.numpy().astype(np.float32)
    out["c1_node_mask"], out["c1_edge_mask"] = nm.numpy(), em.numpy()
    out["c1_cfg"] = np.array(json.dumps(dict(dataset="cata", T=T, eseed=0, nodes=[4] * B, std=1.0)))

    # --- tiny guided / unguided chains, amplified coordinate heads, T=50
    for ci, (name, ds, nodes, amp) in enumerate([("cata_tiny", "cata", [6, 8, 8, 3], False),
                                                 ("hetro_tiny", "hetro", [3, 5, 4], False),
                                                 ("cata_tiny_amp", "cata", [6, 8, 8, 3], True)]):
        F = synth.num_node_features(ds)
        over = dict(diffusion_steps=T, **TINY)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=800 + ci, amplify_coord=amp)
        a, model = build_ref_edm(ds, esd, **over)
        psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=810 + ci,
                                               amplify_coord=amp)
        pa, pred = build_ref_pred(ds, psd, **TINY_P)

        def tf_gap(_in, _nm, _em, _t):
            return -pred(_in, _nm, _em, _t)[:, 1]

        n = torch.tensor(nodes)
        Nn = max(nodes) * (2 if ds != "cata" else 1)
        noise = rng_noise(820 + ci, (T + 2, len(nodes), Nn, 3 + F))
        with InjectNoise(list(noise)):
            x, h, nm, em = ref_sampling.sample_guidance(a, model, tf_gap, n, scale=0.6, std=1.0)
        out[f"{name}_noise"] = noise
        out[f"{name}_x_guided"], out[f"{name}_h_guided"] = x.numpy(), h.numpy().astype(np.float32)
        out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = nm.numpy(), em.numpy()
        a.max_nodes = max(nodes)
        with InjectNoise(list(noise)):
            x, h, nm2, em2 = ref_sampling.sample_pos_edm(a, model, n, std=0.7)
        assert np.array_equal(nm2.numpy(), nm.numpy())
        out[f"{name}_x_unguided"], out[f"{name}_h_unguided"] = x.numpy(), h.numpy().astype(np.float32)
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, eseed=800 + ci, pseed=810 + ci,
                                                      nodes=nodes, amp=amp)))
    save("g7_end_to_end", **out)


def g9_sample_chain():
    """EnVariationalDiffusion.sample_chain (keep_frames trajectory capture), tiny config, injected noise."""
    out = {}
    T, K = 50, 10
    for ci, (name, ds, nodes) in enumerate([("cata", "cata", [6, 8]), ("hetro", "hetro", [3, 5])]):
        F = synth.num_node_features(ds)
        over = dict(diffusion_steps=T, **TINY)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=900 + ci)
        a, model = build_ref_edm(ds, esd, **over)
        nm, em = masks(ds, nodes, None)
        B, N, _ = nm.shape
        noise = rng_noise(920 + ci, (T + 2, B, N, 3 + F))
        with InjectNoise(list(noise)):
            chain = model.sample_chain(B, N, torch.from_numpy(nm), torch.from_numpy(em), None, keep_frames=K, std=0.7)
        out[f"{name}_noise"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = noise, nm, em
        out[f"{name}_chain"] = chain.numpy()
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, K=K, eseed=900 + ci, nodes=nodes, amp=False, std=0.7)))
    save("g9_sample_chain", **out)


def nonlinear_target_torch(pred, t):
    """Golden nonlinear, time-dependent target (numpy twin with its analytic gradient: tests/helpers.py)."""
    return 0.5 * torch.log1p(pred[:, 1] ** 2) + 0.1 * torch.tanh(pred[:, 0]) * pred[:, 3] + t * pred[:, 2]


def g10_nonlinear_target():
    """Guidance with a target that is NOT linear in the predictor outputs: teacher-forced steps and a T=50 chain
    through the reference's sample_guidance (autograd through the closure, en_diffusion.py:899-903)."""
    out = {}
    ds, nodes = "hetro", [3, 5, 4]
    F = synth.num_node_features(ds)
    T = 1000
    esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **TINY), F, seed=900, amplify_coord=True)
    a, model = build_ref_edm(ds, esd, **TINY)
    psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=910, amplify_coord=True)
    pa, pred = build_ref_pred(ds, psd, **TINY_P)

    def tf(_in, _nm, _em, _t):
        return nonlinear_target_torch(pred(_in, _nm, _em, _t), _t.reshape(-1))

    nm, em, z = case_inputs(ds, nodes, None, seed=920, guidance_pad=True)
    B, N, D = z.shape
    tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
    out["z"], out["node_mask"], out["edge_mask"] = z, nm, em
    for s in (0, 500, 999):
        eps = rng_noise(930 + s % 7, (B, N, D))
        st = torch.full((B, 1), s) / T
        tt = (torch.full((B, 1), s) + 1) / T
        out[f"s{s}_eps"] = eps
        for scale in (0.6, 400.0):
            with InjectNoise([eps]), torch.no_grad():
                zg = model.sample_p_zs_given_zt_guidance(st, tt, torch.from_numpy(z), tnm, tem, tf, scale).numpy()
            out[f"s{s}_zs_scale{scale}"] = zg
    out["cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=900, pseed=910, T=T, nodes=nodes)))
    # chain, T=50
    Tc = 50
    over = dict(diffusion_steps=Tc, **TINY)
    esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=940, amplify_coord=False)
    a, model = build_ref_edm(ds, esd, **over)
    psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=950, amplify_coord=False)
    pa, pred = build_ref_pred(ds, psd, **TINY_P)
    noise = rng_noise(960, (Tc + 2, len(nodes), 2 * max(nodes), 3 + F))
    with InjectNoise(list(noise)):
        x, h, nm, em = ref_sampling.sample_guidance(a, model, tf, torch.tensor(nodes), scale=0.6, std=1.0)
    out["chain_noise"], out["chain_x"], out["chain_h"] = noise, x.numpy(), h.numpy().astype(np.float32)
    out["chain_node_mask"], out["chain_edge_mask"] = nm.numpy(), em.numpy()
    out["chain_cfg"] = np.array(json.dumps(dict(dataset=ds, T=Tc, eseed=940, pseed=950, nodes=nodes)))
    save("g10_nonlinear_target", **out)


def _rand_rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q


def synth_cata_molecule(rng, n, tree_prob, jitter, zjit):
    """Ring centres on a triangular lattice (spacing ~2.45): tree-like growth gives cata-condensed shapes, allowing
    lattice neighbours gives cycles (peri shapes, 60-degree triplets) -- a mix of stable / unstable molecules."""
    dirs = [(1, 0), (0, 1), (-1, 1), (-1, 0), (0, -1), (1, -1)]
    occ = [(0, 0)]
    tries = 0
    while len(occ) < n and tries < 1000:
        tries += 1
        a = occ[rng.integers(len(occ))]
        d = dirs[rng.integers(6)]
        c = (a[0] + d[0], a[1] + d[1])
        if c in occ:
            continue
        nb = sum(((c[0] + e[0], c[1] + e[1]) in occ) for e in dirs)
        if nb > 1 and rng.random() < tree_prob:
            continue
        occ.append(c)
    occ = np.array(occ, np.float64)
    d = rng.uniform(2.40, 2.50)
    xy = np.stack([occ[:, 0] + 0.5 * occ[:, 1], occ[:, 1] * np.sqrt(3) / 2], 1) * d
    x = np.concatenate([xy, np.zeros((len(occ), 1))], 1)
    x += rng.standard_normal(x.shape) * jitter
    x[:, 2] += rng.standard_normal(len(occ)) * zjit
    x = (x - x.mean(0)) @ _rand_rot(rng)
    return x.astype(np.float32)


def synth_hetro_molecule(rng, n, tables, jitter, corrupt):
    """Tree growth in the plane: bonded-distance windows and 3-ring angle windows of the reference's tables pick the
    bond length / direction, then jitter.  Orientation nodes (type '.') are appended; `corrupt` breaks them."""
    R = len(tables["rings"]["hetro"]) - 1
    lo, hi, a3 = tables["dist_lo"]["hetro"], tables["dist_hi"]["hetro"], tables["a3"]["hetro"]
    w = np.array([6, 1, 1, 1, 1, 1, 1, 1, 0.3, 1, 0.5])
    types = [int(rng.choice(R, p=w / w.sum()))]
    pos = [np.zeros(2)]
    heading = [rng.uniform(0, 2 * np.pi)]  # direction towards the parent (or arbitrary for the root)
    tries = 0
    while len(pos) < n and tries < 2000:
        tries += 1
        p = int(rng.integers(len(pos)))
        t = int(rng.choice(R, p=w / w.sum()))
        if hi[types[p]][t] <= 0:
            continue
        d = rng.uniform(lo[types[p]][t], hi[types[p]][t])
        wins = a3[types[p]] or [[120.0, 120.0]]
        win = wins[rng.integers(len(wins))]
        ang = np.deg2rad(rng.uniform(win[0], win[1])) * rng.choice([-1, 1])
        th = heading[p] + ang
        c = pos[p] + d * np.array([np.cos(th), np.sin(th)])
        if min(np.linalg.norm(c - q) for q in pos) < 1.3 and rng.random() < 0.9:
            continue
        pos.append(c)
        types.append(t)
        heading.append(th + np.pi)
    k = len(pos)
    x = np.concatenate([np.array(pos), np.zeros((k, 1))], 1)
    x += rng.standard_normal(x.shape) * jitter
    orient = x + rng.standard_normal(x.shape) * 0.5
    x = np.concatenate([x, orient], 0)
    x = (x - x.mean(0)) @ _rand_rot(rng)
    ty = np.array(types + [R] * k, np.int64)
    if corrupt == 1:
        ty[k + rng.integers(k)] = 0          # an orientation slot holds a real ring type
    elif corrupt == 2:
        ty[rng.integers(k)] = R              # a ring slot holds the orientation type
    return x.astype(np.float32), ty


def g11_stability():
    """Graph-of-rings stability check (analyze/analyze.py:50-100) on seeded synthetic molecules + a few degenerate
    ones; stores the reference's flags, distance matrix, adjacency and the sorted 3-ring / 4-ring angle lists."""
    from analyze import analyze as ref_an
    from utils import helpers as ref_h
    tables = json.load(open(os.path.join(ROOT, "gaudi_amd", "data", "ring_tables.json")))
    rng = np.random.default_rng(1100)
    out = {}
    for ds in ("cata", "hetro"):
        mols = []
        if ds == "cata":
            for i in range(260):
                n = int(rng.integers(1, 12))
                jit = [0.0, 0.01, 0.03, 0.08, 0.2][i % 5]
                zj = [0.0, 0.02, 0.3, 0.8][(i // 5) % 4]
                x = synth_cata_molecule(rng, n, tree_prob=[1.0, 0.9, 0.5][i % 3], jitter=jit, zjit=zj)
                mols.append((x, np.zeros(len(x), np.int64)))
            # degenerate: two far-apart fragments, two coincident rings, an exactly straight chain
            mols.append((np.array([[0, 0, 0], [2.45, 0, 0], [9, 0, 0], [11.45, 0, 0]], np.float32), np.zeros(4, np.int64)))
            mols.append((np.array([[0, 0, 0], [2.45, 0, 0], [2.45, 0.1, 0]], np.float32), np.zeros(3, np.int64)))
            mols.append((np.array([[2.45 * i, 0, 0] for i in range(5)], np.float32), np.zeros(5, np.int64)))
        else:
            for i in range(260):
                n = int(rng.integers(2, 11))
                x, ty = synth_hetro_molecule(rng, n, tables, jitter=[0.0, 0.01, 0.03, 0.1][i % 4],
                                             corrupt=[0, 0, 0, 0, 0, 0, 1, 2][i % 8])
                mols.append((x, ty))
        M = len(mols)
        NM = max(len(x) for x, _ in mols)
        X = np.zeros((M, NM, 3), np.float32)
        TY = -np.ones((M, NM), np.int64)
        NN = np.zeros(M, np.int64)
        FL = np.zeros((M, 5), np.uint8)
        DIST = np.zeros((M, NM, NM), np.float32)
        ADJ = np.zeros((M, NM, NM), np.float32)
        A3 = np.full((M, 128), np.nan, np.float32)
        A3T = -np.ones((M, 128), np.int64)
        A4 = np.full((M, 256), np.nan, np.float32)
        CNT = np.zeros((M, 2), np.int64)
        for m, (x, ty) in enumerate(mols):
            k = len(x)
            X[m, :k], TY[m, :k], NN[m] = x, ty, k
            res = ref_an.check_stability(torch.from_numpy(x.copy()), torch.from_numpy(ty.copy()), dataset=ds)
            FL[m] = [res[key] for key in ("orientation_nodes", "dist_stable", "connected", "angels3", "angels4")]
            nr = k if ds == "cata" else k // 2
            if res["orientation_nodes"]:
                dist, adj = ref_h.positions2adj(torch.from_numpy(x[None, :nr].copy()), torch.from_numpy(ty[None, :nr].copy()),
                                                0.1, dataset=ds)
                DIST[m, :nr, :nr], ADJ[m, :nr, :nr] = dist[0].numpy(), adj[0].numpy()
                if res["connected"]:
                    a3, a4 = ref_an.get_angels(torch.from_numpy(x[None, :nr].copy()), torch.from_numpy(ty[None, :nr].copy()),
                                               adj, dataset=ds)
                    rl = tables["rings"][ds]
                    pairs = [(rl.index(sym), float(a)) for sym, a in a3]
                    for i, (t, a) in enumerate(pairs):
                        A3T[m, i], A3[m, i] = t, a
                    for i, a in enumerate(float(a) for _, a in a4):
                        A4[m, i] = a
                    CNT[m] = len(a3), len(a4)
        out.update({f"{ds}_x": X, f"{ds}_types": TY, f"{ds}_n": NN, f"{ds}_flags": FL, f"{ds}_dist": DIST,
                    f"{ds}_adj": ADJ, f"{ds}_a3": A3, f"{ds}_a3_type": A3T, f"{ds}_a4": A4, f"{ds}_counts": CNT})
        print(ds, "flag rates", FL.mean(0), "all-stable", FL.all(1).mean())
    save("g11_stability", **out)


def g12_ring_count_sampler():
    """DistributionRings (models_edm.py:21-58): seeded draws and log-probabilities of the reference."""
    out = {}
    for ds in ("cata", "hetro"):
        d = models_edm.DistributionRings(ds)
        torch.manual_seed(1234)
        s = d.sample(2000)
        out[f"{ds}_sample"] = s.numpy()
        out[f"{ds}_log_prob"] = d.log_prob(s[:64]).numpy()
        out[f"{ds}_n_nodes"] = d.n_nodes.numpy()
        out[f"{ds}_prob"] = d.prob.numpy()
    save("g12_ring_count_sampler", **out)


def g13_noised_predictor():
    """sample_edm_t + compute_loss (cond_prediction/train_cond_predictor.py:47-81): forward-noised inputs, predictor
    outputs at the noise level and the per-target absolute errors, for fixed and per-sample t."""
    from cond_prediction import train_cond_predictor as tcp
    out = {}
    for ci, (name, ds, nodes) in enumerate([("cata", "cata", [4, 11, 7, 1, 11]), ("hetro", "hetro", [3, 9, 6, 10])]):
        F = synth.num_node_features(ds)
        eargs = synth.edm_args(dataset=ds, **TINY)
        esd = synth.synth_edm_state_dict(eargs, F, seed=1300 + ci)
        a, model = build_ref_edm(ds, esd, **TINY)
        psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=1310 + ci, amplify_coord=True)
        pa, pred = build_ref_pred(ds, psd, **TINY_P)
        nm, em, z = case_inputs(ds, nodes, None, seed=1320 + ci, guidance_pad=(ds != "cata"))
        B, N, D = z.shape
        rng = np.random.default_rng(1330 + ci)
        x = (z[:, :, :3] * 3.0).astype(np.float32)                      # un-normalised, masked, mean-free positions
        cls = rng.integers(0, F, (B, N))
        h = (np.eye(F, dtype=np.float32)[cls] * nm).astype(np.float32)   # one-hot ring types
        y = rng.standard_normal((B, 5)).astype(np.float32)
        out[f"{name}_x"], out[f"{name}_h"], out[f"{name}_y"] = x, h, y
        out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = nm, em
        tx, th, tnm, tem = (torch.from_numpy(v) for v in (x, h, nm, em))
        T = a.diffusion_steps
        for tag, t_int in (("t0", np.zeros(B)), ("t500", np.full(B, 500.0)), ("tT", np.full(B, float(T))),
                           ("tmix", rng.integers(0, T + 1, B).astype(np.float64))):
            eps = rng_noise(1340 + ci + len(tag), (B, N, D))
            t = torch.from_numpy((t_int / T).astype(np.float32)).view(B, 1)
            with InjectNoise([eps]), torch.no_grad():
                zt = tcp.sample_edm_t(tx, th, model, t, tnm)
                p = pred(zt, tnm, tem.view(B, N * N), t)
            out[f"{name}_{tag}_t_int"] = t_int.astype(np.int32)
            out[f"{name}_{tag}_eps"], out[f"{name}_{tag}_zt"], out[f"{name}_{tag}_pred"] = eps, zt.numpy(), p.numpy()
            if tag == "t500":   # the whole compute_loss with t_fix (same noise -> same z_t)
                with InjectNoise([eps]), torch.no_grad():
                    loss, err = tcp.compute_loss(pred, tx, th, tnm, tem, torch.from_numpy(y), model, a, t_fix=500)
                out[f"{name}_t500_loss"], out[f"{name}_t500_err"] = np.float32(loss.item()), err.numpy()
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=1300 + ci, pseed=1310 + ci, T=T)))
    save("g13_noised_predictor", **out)


def g14_long_chains():
    """T = 1000 chains at the DEFAULT architectures (the bench's C2 / C3 weights: EDM seed 0, predictor seed 1), B = 8,
    N = 11, injected noise regenerated from a seed (checksummed).  Unguided final (x, h); guided final (x, h) plus the
    reference's own (z_t, z_s) at 26 points of its trajectory for teacher-forced step parity along the real chain."""
    out = {}
    T = 1000
    nodes = [11, 11, 11, 7, 4, 11, 9, 11]
    B, N, D = len(nodes), 11, 4
    over = dict(diffusion_steps=T)
    esd = synth.synth_edm_state_dict(synth.edm_args(**over), 1, seed=0)
    a, model = build_ref_edm("cata", esd, **over)
    psd = synth.synth_predictor_state_dict(synth.pred_args(), 1, 5, seed=1)
    pa, pred = build_ref_pred("cata", psd)
    noise = rng_noise(1400, (T + 2, B, N, D))
    out["noise_seed"] = np.int64(1400)
    out["noise_checksum"] = np.array([np.float64(noise.astype(np.float64).sum()), np.float64(np.abs(noise).astype(np.float64).sum()),
                                      np.float64(noise[17, 3, 5, 2]), np.float64(noise[T + 1, 7, 10, 3])])
    a.max_nodes = N
    import time
    t0 = time.time()
    with InjectNoise(list(noise)):
        x, h, nm, em = ref_sampling.sample_pos_edm(a, model, torch.tensor(nodes), std=1.0)
    print(f"g14 unguided T=1000: {time.time() - t0:.0f} s")
    out["unguided_x"], out["unguided_h"] = x.numpy(), h.numpy().astype(np.float32)
    out["node_mask"], out["edge_mask"] = nm.numpy(), em.numpy()

    def tf_gap(_in, _nm, _em, _t):
        return -pred(_in, _nm, _em, _t)[:, 1]

    pts = [999, 990, 950, 900, 850, 800, 750, 700, 650, 600, 550, 500, 450, 400, 350, 300, 250, 200, 150, 100, 50, 25, 10, 5, 1, 0]
    rec = {}
    orig = model.sample_p_zs_given_zt_guidance

    def wrapped(s, t, zt, node_mask, edge_mask, target_function, scale, fix_noise=False):
        zs = orig(s, t, zt, node_mask, edge_mask, target_function, scale, fix_noise=fix_noise)
        si = int(round(float(s[0, 0]) * T))
        if si in pts:
            rec[si] = (zt.detach().numpy().copy(), zs.detach().numpy().copy())
        return zs

    model.sample_p_zs_given_zt_guidance = wrapped
    t0 = time.time()
    with InjectNoise(list(noise)):
        x, h, nm2, em2 = ref_sampling.sample_guidance(a, model, tf_gap, torch.tensor(nodes), scale=0.6, std=1.0)
    print(f"g14 guided T=1000: {time.time() - t0:.0f} s")
    model.sample_p_zs_given_zt_guidance = orig
    assert np.array_equal(nm2.numpy(), nm.numpy()) and sorted(rec) == sorted(pts)
    out["guided_x"], out["guided_h"] = x.numpy(), h.numpy().astype(np.float32)
    # the reference's OWN rounding sensitivity on exactly these chains: the same model and noise in float64
    model.double()
    pred.double()
    with InjectNoise(list(noise), torch.float64):
        x64, _, _, _ = ref_sampling.sample_pos_edm(a, model, torch.tensor(nodes), std=1.0)
    with InjectNoise(list(noise), torch.float64):
        xg64, _, _, _ = ref_sampling.sample_guidance(a, model, tf_gap, torch.tensor(nodes), scale=0.6, std=1.0)
    out["unguided_x_fp64"], out["guided_x_fp64"] = x64.numpy(), xg64.numpy()
    mx = lambda p, q: float(np.abs(p.astype(np.float64) - q.astype(np.float64)).max() / np.abs(q).max())
    out["spread_unguided"] = np.float64(mx(out["unguided_x"], out["unguided_x_fp64"]))
    out["spread_guided"] = np.float64(mx(out["guided_x"], out["guided_x_fp64"]))
    print("g14 reference fp32-vs-fp64 spread: unguided %.2e guided %.2e" % (out["spread_unguided"], out["spread_guided"]))
    out["traj_s"] = np.array(pts, np.int32)
    out["traj_zt"] = np.stack([rec[s][0] for s in pts])
    out["traj_zs"] = np.stack([rec[s][1] for s in pts])
    out["cfg"] = np.array(json.dumps(dict(dataset="cata", T=T, eseed=0, pseed=1, nodes=nodes, scale=0.6, std=1.0)))
    save("g14_long_chains", **out)


def g15_nan_scrub():
    """NaN handling of the reverse steps (edm/egnn/models.py:138-141, en_diffusion.py:881,933-934): a NaN planted in the
    last block's coordinate head makes phi's velocity NaN (scrubbed to 0 inside phi); a NaN planted in the predictor's
    readout makes the guidance gradient NaN (z_s scrubbed to 0 at the end of the guided step)."""
    out = {}
    T = 1000
    ds, nodes = "cata", [4, 11, 7, 11]
    F = 1
    eargs = synth.edm_args(dataset=ds, **TINY)
    esd = synth.synth_edm_state_dict(eargs, F, seed=1500, amplify_coord=True)
    pargs = synth.pred_args(dataset=ds, **TINY_P)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1510, amplify_coord=True)
    nm, em, z = case_inputs(ds, nodes, None, seed=1520, guidance_pad=True)
    B, N, D = z.shape
    tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
    out["z"], out["node_mask"], out["edge_mask"] = z, nm, em
    esd_bad = {k: v.copy() for k, v in esd.items()}
    key_e = "dynamics.egnn.e_block_1.gcl_equiv.coord_mlp.4.weight"
    esd_bad[key_e][0, 3] = np.nan
    psd_bad = {k: v.copy() for k, v in psd.items()}
    key_p = "egnn.embedding_out.weight"
    psd_bad[key_p][1, 5] = np.nan
    out["edm_poison_key"], out["edm_poison_idx"] = np.array(key_e), np.array([0, 3])
    out["pred_poison_key"], out["pred_poison_idx"] = np.array(key_p), np.array([1, 5])
    a, model_bad = build_ref_edm(ds, esd_bad, **TINY)
    a2, model_ok = build_ref_edm(ds, esd, **TINY)
    pa, pred_ok = build_ref_pred(ds, psd, **TINY_P)
    pb, pred_bad = build_ref_pred(ds, psd_bad, **TINY_P)
    import contextlib, io
    for s in (999, 500, 0):
        eps = rng_noise(1530 + s % 7, (B, N, D))
        st = torch.full((B, 1), s) / T
        tt = (torch.full((B, 1), s) + 1) / T
        out[f"s{s}_eps"] = eps
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            e = model_bad.phi(torch.from_numpy(z), tt, tnm, tem, None).numpy()
            with InjectNoise([eps]):
                zu = model_bad.sample_p_zs_given_zt(st, tt, torch.from_numpy(z), tnm, tem, None).numpy()
            with InjectNoise([eps]):
                zg = model_bad.sample_p_zs_given_zt_guidance(
                    st, tt, torch.from_numpy(z), tnm, tem, lambda i, n, m, t: -pred_ok(i, n, m, t)[:, 1], 0.6).numpy()
            with InjectNoise([eps]):
                zp = model_ok.sample_p_zs_given_zt_guidance(
                    st, tt, torch.from_numpy(z), tnm, tem, lambda i, n, m, t: -pred_bad(i, n, m, t)[:, 1], 0.6).numpy()
        assert np.isfinite(e).all() and np.isfinite(zu).all() and np.isfinite(zg).all() and np.isfinite(zp).all()
        out[f"s{s}_phi_edm_poisoned"] = e
        out[f"s{s}_zs_unguided_edm_poisoned"] = zu
        out[f"s{s}_zs_guided_edm_poisoned"] = zg
        out[f"s{s}_zs_guided_pred_poisoned"] = zp
    out["cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=1500, pseed=1510, T=T, nodes=nodes)))
    save("g15_nan_scrub", **out)


def g17_nan_in_edge_gemm_matrix():
    """NaN planted in a matrix of the EDGE-level GEMMs (edge_mlp.2.weight of an EDM GCL; edge_mlp.2.weight and
    coord_mlp.0.weight of a predictor layer) -- the matrices the default kernels stream as three bf16 pieces.  The EDM's
    h output is NaN (only the velocity is scrubbed, edm/egnn/models.py:138-141), the guided step zeroes eps_hat
    (en_diffusion.py:881) and stays finite; a poisoned predictor gives a NaN gradient and z_s is scrubbed to zeros (:933-934)."""
    out = {}
    T = 1000
    ds, nodes = "cata", [4, 11, 7, 11]
    F = 1
    eargs = synth.edm_args(dataset=ds, **TINY)
    esd = synth.synth_edm_state_dict(eargs, F, seed=1700, amplify_coord=True)
    pargs = synth.pred_args(dataset=ds, **TINY_P)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1710, amplify_coord=True)
    nm, em, z = case_inputs(ds, nodes, None, seed=1720, guidance_pad=True)
    B, N, D = z.shape
    tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
    out["z"], out["node_mask"], out["edge_mask"] = z, nm, em
    poison = dict(edm=("dynamics.egnn.e_block_1.gcl_0.edge_mlp.2.weight", [5, 17]),
                  pred_w2=("egnn.gcl_1.edge_mlp.2.weight", [30, 2]),
                  pred_wc1=("egnn.gcl_0.coord_mlp.0.weight", [7, 33]))
    bad = {}
    for name, (key, idx) in poison.items():
        src = esd if name == "edm" else psd
        d = {k: v.copy() for k, v in src.items()}
        d[key][tuple(idx)] = np.nan
        bad[name] = d
        out[f"{name}_key"], out[f"{name}_idx"] = np.array(key), np.array(idx)
    a, model_bad = build_ref_edm(ds, bad["edm"], **TINY)
    a2, model_ok = build_ref_edm(ds, esd, **TINY)
    pa, pred_ok = build_ref_pred(ds, psd, **TINY_P)
    preds_bad = {k: build_ref_pred(ds, bad[k], **TINY_P)[1] for k in ("pred_w2", "pred_wc1")}
    import contextlib, io
    for s in (999, 500, 0):
        eps = rng_noise(1730 + s % 7, (B, N, D))
        st = torch.full((B, 1), s) / T
        tt = (torch.full((B, 1), s) + 1) / T
        out[f"s{s}_eps"] = eps
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            e = model_bad.phi(torch.from_numpy(z), tt, tnm, tem, None).numpy()
            with InjectNoise([eps]):
                zg = model_bad.sample_p_zs_given_zt_guidance(
                    st, tt, torch.from_numpy(z), tnm, tem, lambda i, n, m, t: -pred_ok(i, n, m, t)[:, 1], 0.6).numpy()
            zp = {}
            for k, pb in preds_bad.items():
                with InjectNoise([eps]):
                    zp[k] = model_ok.sample_p_zs_given_zt_guidance(
                        st, tt, torch.from_numpy(z), tnm, tem, lambda i, n, m, t, pb=pb: -pb(i, n, m, t)[:, 1], 0.6).numpy()
        assert np.isfinite(e[:, :, :3]).all() and np.isnan(e[:, :, 3:][nm[:, :, 0] != 0]).all()
        assert np.isfinite(zg).all() and all(np.isfinite(v).all() for v in zp.values())
        out[f"s{s}_phi_edm_poisoned"] = e
        out[f"s{s}_zs_guided_edm_poisoned"] = zg
        for k, v in zp.items():
            out[f"s{s}_zs_guided_{k}_poisoned"] = v
    out["cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=1700, pseed=1710, T=T, nodes=nodes)))
    save("g17_nan_edge_matrix", **out)


def g18_large_molecules():
    """Molecules beyond the LDS limit of the library's resident kernels: hetero 20 rings (N = 40 graph nodes with the
    orientation nodes: BASELINE config 4's "ring count 6-20" read literally; sampling_edm.py:172-209 has no cap) and a
    14-ring one in the same batch, DEFAULT architectures: phi, predictor + input gradient, one teacher-forced guided step."""
    out = {}
    T = 1000
    ds, nodes = "hetro", [20, 14]
    F = synth.num_node_features(ds)
    eargs = synth.edm_args(dataset=ds)
    esd = synth.synth_edm_state_dict(eargs, F, seed=1800, amplify_coord=True)
    pargs = synth.pred_args(dataset=ds)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1810, amplify_coord=True)
    a, model = build_ref_edm(ds, esd)
    pa, pred = build_ref_pred(ds, psd)
    nm, em, z = case_inputs(ds, nodes, None, seed=1820, guidance_pad=True)
    B, N, D = z.shape
    assert N == 40
    tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
    out["z"], out["node_mask"], out["edge_mask"] = z, nm, em
    s = 600
    st = torch.full((B, 1), s) / T
    tt = (torch.full((B, 1), s) + 1) / T
    eps = rng_noise(1830, (B, N, D))
    out["eps"] = eps
    with torch.no_grad():
        out["phi"] = model.phi(torch.from_numpy(z), tt, tnm, tem, None).numpy()
    zt = torch.from_numpy(z).requires_grad_()
    p = pred(zt, tnm, tem, tt)
    out["pred"] = p.detach().numpy()
    out["grad_gap"] = torch.autograd.grad((0.6 * -p[:, 1]).sum(), zt)[0].numpy()
    with InjectNoise([eps]), torch.no_grad():
        out["zs_unguided"] = model.sample_p_zs_given_zt(st, tt, torch.from_numpy(z), tnm, tem, None).numpy()
    with InjectNoise([eps]), torch.no_grad():
        out["zs_guided"] = model.sample_p_zs_given_zt_guidance(st, tt, torch.from_numpy(z), tnm, tem,
                                                               lambda i, n, m, t: -pred(i, n, m, t)[:, 1], 0.6).numpy()
    out["cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=1800, pseed=1810, T=T, s=s, nodes=nodes)))
    save("g18_large_molecules", **out)


def g19_amplified_default_steps():
    """Reference-held anchor for the ill-conditioned case: DEFAULT architectures with amplified coordinate heads (cata C3
    shape and hetero C4 shape), teacher-forced unguided / guided steps at s = 999, 400, 0 -- run by the reference in fp32 AND
    in float64 (model.double()), so that a kernel can be held to 1e-4 against the reference's own float64 result where
    the two fp32 evaluations differ from each other by about that much."""
    out = {}
    T = 1000
    for ci, (name, ds, nodes) in enumerate([("cata", "cata", [11, 11, 7, 11, 4, 9]), ("hetro", "hetro", [10, 3, 7, 5])]):
        F = synth.num_node_features(ds)
        eargs = synth.edm_args(dataset=ds)
        esd = synth.synth_edm_state_dict(eargs, F, seed=1900 + ci, amplify_coord=True)
        pargs = synth.pred_args(dataset=ds)
        psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1910 + ci, amplify_coord=True)
        a, model = build_ref_edm(ds, esd)
        pa, pred = build_ref_pred(ds, psd)
        nm, em, z = case_inputs(ds, nodes, None, seed=1920 + ci, guidance_pad=True)
        B, N, D = z.shape
        tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
        wv = torch.tensor([0., -1., 0., 0., 0.] if ds == "cata" else [3., 0., 1., 1., 0.])
        out[f"{name}_z"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"], out[f"{name}_w"] = z, nm, em, wv.numpy()
        for dt, tag in ((torch.float32, "fp32"), (torch.float64, "fp64")):
            if dt == torch.float64:
                model.double()
                pred.double()
            zt_, nm_, em_ = torch.from_numpy(z).to(dt), tnm.to(dt), tem.to(dt)
            tf = lambda i, n, m, t: (pred(i, n, m, t) * wv.to(dt)).sum(-1)
            for s in (999, 400, 0):
                eps = rng_noise(1930 + 10 * ci + s % 7, (B, N, D))
                st = (torch.full((B, 1), s) / T).to(dt)
                tt = ((torch.full((B, 1), s) + 1) / T).to(dt)
                out[f"{name}_s{s}_eps"] = eps
                with InjectNoise([eps], dt), torch.no_grad():
                    out[f"{name}_s{s}_zs_unguided_{tag}"] = model.sample_p_zs_given_zt(st, tt, zt_, nm_, em_, None).numpy()
                with InjectNoise([eps], dt), torch.no_grad():
                    out[f"{name}_s{s}_zs_guided_{tag}"] = model.sample_p_zs_given_zt_guidance(st, tt, zt_, nm_, em_, tf, 0.6).numpy()
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, eseed=1900 + ci, pseed=1910 + ci, T=T, nodes=nodes)))
    save("g19_amplified_default_steps", **out)


def g16_fix_noise():
    """fix_noise=True (en_diffusion.py:562-566,972-978,1022-1028): ONE raw draw [1,N,3+F] per call is broadcast over the
    batch and masked / mean-centred per molecule.  Tiny config, T = 50, unguided and guided."""
    out = {}
    T = 50
    ds, nodes = "cata", [6, 8, 8, 3]
    F = 1
    over = dict(diffusion_steps=T, **TINY)
    esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=1600)
    a, model = build_ref_edm(ds, esd, **over)
    psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=1610)
    pa, pred = build_ref_pred(ds, psd, **TINY_P)
    nm, em = masks(ds, nodes, None)
    B, N, _ = nm.shape
    noise = rng_noise(1620, (T + 2, 1, N, 3 + F))
    tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
    with InjectNoise(list(noise)):
        x, h = model.sample(B, N, tnm, tem, fix_noise=True, std=0.7)
    out["x_unguided"], out["h_unguided"] = x.numpy(), h["categorical"].numpy().astype(np.float32)
    with InjectNoise(list(noise)):
        x, h = model.sample_guidance(B, lambda i, n, m, t: -pred(i, n, m, t)[:, 1], tnm, tem, scale=0.6, fix_noise=True, std=1.0)
    out["x_guided"], out["h_guided"] = x.numpy(), h["categorical"].numpy().astype(np.float32)
    out["noise"], out["node_mask"], out["edge_mask"] = noise, nm, em
    out["cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, eseed=1600, pseed=1610, nodes=nodes, amp=False)))
    save("g16_fix_noise", **out)


def g20_cosine_and_mean():
    """The two EDM modes SURVEY 2 allowed to refuse and VERDICT r3 asked to accept: diffusion_noise_schedule='cosine'
    (en_diffusion.py:64-81,196-197) and aggregation_method='mean' (egnn_new.py:403-421: the norm counts EVERY edge of the dense
    list, masked or not, i.e. the padded node count N).  gamma / coefficient tables, phi with 'mean', and guided T=50 chains
    with both switched on, tiny configs, injected noise."""
    out = {}
    for T in (50, 1000):
        sd = synth.synth_edm_state_dict(synth.edm_args(nf=8, n_layers=1), 1, seed=0)
        # (the reference refuses 'cosine' with the default normalize_factors [3, 4, 10]: sigma_0 = 0.041 is too large for a
        # one-hot scaled by 1/4, en_diffusion.py:336-349 -- a cosine model needs smaller ones)
        a, model = build_ref_edm("cata", sd, nf=8, n_layers=1, diffusion_steps=T, diffusion_noise_schedule="cosine",
                                 normalize_factors=[1, 2, 2])
        out[f"gamma_T{T}"] = model.gamma.gamma.numpy().copy()
        rows = []
        for s in ([0, 1, T // 2, T - 2, T - 1]):
            st = torch.full((1, 1), s) / T
            tt = (torch.full((1, 1), s) + 1) / T
            gs, gt = model.gamma(st), model.gamma(tt)
            zt = torch.zeros(1, 1, 1)
            s2, s_ts, a_ts = model.sigma_and_alpha_t_given_s(gt, gs, zt)
            sig_s, sig_t = model.sigma(gs, zt), model.sigma(gt, zt)
            rows.append([s, a_ts.item(), s2.item(), (s2 / a_ts / sig_t).item(), (s_ts * sig_s / sig_t).item(),
                         sig_s.item(), sig_t.item(), tt.item()])
        out[f"coef_T{T}"] = np.array(rows, dtype=np.float64)
    T = 50
    for ci, (name, ds, nodes, mx) in enumerate([("cata", "cata", [4, 11, 7, 2, 11], 11), ("hetro", "hetro", [3, 5, 10, 7], 10)]):
        F = synth.num_node_features(ds)
        over = dict(diffusion_steps=T, aggregation_method="mean", diffusion_noise_schedule="cosine", normalize_factors=[1, 2, 2], **TINY)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=2000 + ci, amplify_coord=True)
        a, model = build_ref_edm(ds, esd, **over)
        psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=2010 + ci, amplify_coord=True)
        pa, pred = build_ref_pred(ds, psd, **TINY_P)
        nm, em, z = case_inputs(ds, nodes, mx, seed=2020 + ci)
        B = z.shape[0]
        t = np.linspace(0.05, 0.95, B).astype(np.float32).reshape(B, 1)
        with torch.no_grad():
            eps = model.phi(torch.from_numpy(z), torch.from_numpy(t), torch.from_numpy(nm), torch.from_numpy(em), None).numpy()
        out[f"{name}_z"], out[f"{name}_t"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = z, t, nm, em
        out[f"{name}_eps"] = eps

        # the chains run on DEFAULT-init weights (a free-running chain through amplified heads is not a 1e-4 comparison:
        # BASELINE.md section 2), the phi cases above on amplified ones
        esd_c = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=2040 + ci)
        a_c, model_c = build_ref_edm(ds, esd_c, **over)
        psd_c = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=2050 + ci)
        pa_c, pred_c = build_ref_pred(ds, psd_c, **TINY_P)

        def tf_gap(_in, _nm, _em, _t):
            return -pred_c(_in, _nm, _em, _t)[:, 1]

        n = torch.tensor(nodes)
        Nn = max(nodes) * (2 if ds != "cata" else 1)
        noise = rng_noise(2030 + ci, (T + 2, len(nodes), Nn, 3 + F))
        with InjectNoise(list(noise)):
            x, h, nm2, em2 = ref_sampling.sample_guidance(a_c, model_c, tf_gap, n, scale=0.6, std=1.0)
        out[f"{name}_noise"], out[f"{name}_x_guided"], out[f"{name}_h_guided"] = noise, x.numpy(), h.numpy().astype(np.float32)
        out[f"{name}_chain_node_mask"], out[f"{name}_chain_edge_mask"] = nm2.numpy(), em2.numpy()
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, eseed=2000 + ci, pseed=2010 + ci, chain_eseed=2040 + ci,
                                                      chain_pseed=2050 + ci, nodes=nodes, amp=True,
                                                      over=dict(TINY, aggregation_method="mean", diffusion_noise_schedule="cosine",
                                                                normalize_factors=[1, 2, 2]))))
    save("g20_cosine_and_mean", **out)


def g22_sin_embedding():
    """`sin_embedding=True` (utils/args_edm.py:33, models_edm.py:78; egnn_new.py:269-273,302-303,217-218,378-391): the two scalar
    edge features r, d0 become 2 x 12 sinusoids of their square roots, the first Linear of every edge / coordinate MLP takes
    2 H + 24 inputs.  phi (tiny widths, amplified heads, cata and hetro; default widths, default init), a teacher-forced
    unguided and guided step at the default widths, run by the reference in fp32 AND float64 (the highest frequency multiplies
    sqrt(r) by 429: the embedding is ill-conditioned in fp32 by construction -- the float64 run says by how much)."""
    out = {}
    from edm.egnn.egnn_new import SinusoidsEmbeddingNew
    out["frequencies"] = SinusoidsEmbeddingNew().frequencies.numpy().copy()
    for ci, (name, ds, nodes, mx, over) in enumerate([("cata_tiny", "cata", [4, 11, 7, 2, 11], 11, dict(TINY)),
                                                      ("hetro_tiny", "hetro", [3, 5, 10, 7], 10, dict(TINY)),
                                                      ("cata_default", "cata", [11, 9, 11], 11, {})]):
        F = synth.num_node_features(ds)
        T = 1000
        over = dict(over, sin_embedding=True, diffusion_steps=T)
        amp = "tiny" in name
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=2200 + ci, amplify_coord=amp)
        a, model = build_ref_edm(ds, esd, **over)
        nm, em, z = case_inputs(ds, nodes, mx, seed=2220 + ci)
        B = z.shape[0]
        t = np.linspace(0.05, 0.95, B).astype(np.float32).reshape(B, 1)
        with torch.no_grad():
            eps = model.phi(torch.from_numpy(z), torch.from_numpy(t), torch.from_numpy(nm), torch.from_numpy(em), None).numpy()
            m64 = copy.deepcopy(model).double()
            eps64 = m64.phi(torch.from_numpy(z).double(), torch.from_numpy(t).double(), torch.from_numpy(nm).double(),
                            torch.from_numpy(em).double(), None).numpy()
        out[f"{name}_z"], out[f"{name}_t"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = z, t, nm, em
        out[f"{name}_eps"], out[f"{name}_eps64"] = eps, eps64
        cfgd = dict(dataset=ds, T=T, eseed=2200 + ci, nodes=nodes, amp=amp, over={k: v for k, v in over.items() if k != "diffusion_steps"})
        if name == "cata_default":
            psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds), F, 5, seed=2210 + ci)
            pa, pred = build_ref_pred(ds, psd)

            def tf_gap(_in, _nm, _em, _t):
                return -pred(_in, _nm, _em, _t)[:, 1]

            eps_n = rng_noise(2230 + ci, z.shape)
            for sidx in (999, 400, 0):
                s_t = torch.full((B, 1), sidx / T)
                t_t = torch.full((B, 1), (sidx + 1) / T)
                with InjectNoise([eps_n]):
                    zs_u = model.sample_p_zs_given_zt(s_t, t_t, torch.from_numpy(z), torch.from_numpy(nm), torch.from_numpy(em), None)
                with InjectNoise([eps_n]):
                    zs_g = model.sample_p_zs_given_zt_guidance(s_t, t_t, torch.from_numpy(z), torch.from_numpy(nm), torch.from_numpy(em),
                                                               tf_gap, 0.6)
                out[f"{name}_zs_unguided_s{sidx}"], out[f"{name}_zs_guided_s{sidx}"] = zs_u.numpy(), zs_g.detach().numpy()
            out[f"{name}_step_noise"] = eps_n
            cfgd["pseed"] = 2210 + ci
        out[f"{name}_cfg"] = np.array(json.dumps(cfgd))
    save("g22_sin_embedding", **out)


def g23_attention_tanh_flags():
    """`--attention False` / `--tanh False` (utils/args_edm.py:29-30, cond_prediction/prediction_args.py:44-45; egnn_new.py:53-57,117-125,
    egnn_predictor/gcl.py): the edge gate and the tanh bound of the coordinate update are constructor switches of BOTH networks.
    Every combination other than the default (True, True), on both networks at once: phi, predictor + input gradient, a
    teacher-forced unguided and guided step -- tiny widths (amplified heads, cata and hetro) and the default widths (cata)."""
    out = {}
    T = 1000
    cases = []
    for att, th in ((False, True), (True, False), (False, False)):
        tag = f"att{int(att)}_tanh{int(th)}"
        cases.append((f"cata_tiny_{tag}", "cata", [4, 11, 7, 2, 11], 11, dict(TINY), dict(TINY_P), True, att, th))
        cases.append((f"hetro_tiny_{tag}", "hetro", [3, 5, 10, 7], 10, dict(TINY), dict(TINY_P), True, att, th))
    cases.append(("cata_default_att0_tanh0", "cata", [11, 9, 11], 11, {}, {}, False, False, False))
    for ci, (name, ds, nodes, mx, over_e, over_p, amp, att, th) in enumerate(cases):
        F = synth.num_node_features(ds)
        over_e = dict(over_e, attention=att, tanh=th, diffusion_steps=T)
        over_p = dict(over_p, attention=att, tanh=th)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over_e), F, seed=2300 + ci, amplify_coord=amp)
        psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **over_p), F, 5, seed=2350 + ci, amplify_coord=amp)
        a, model = build_ref_edm(ds, esd, **over_e)
        pa, pred = build_ref_pred(ds, psd, **over_p)
        assert ("dynamics.egnn.e_block_0.gcl_0.att_mlp.0.weight" in esd) == att
        nm, em, z = case_inputs(ds, nodes, mx, seed=2320 + ci)
        B = z.shape[0]
        t = np.linspace(0.05, 0.95, B).astype(np.float32).reshape(B, 1)
        tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
        with torch.no_grad():
            out[f"{name}_eps"] = model.phi(torch.from_numpy(z), torch.from_numpy(t), tnm, tem, None).numpy()
        zt = torch.from_numpy(z).requires_grad_()
        p = pred(zt, tnm, tem, torch.from_numpy(t))
        out[f"{name}_pred"] = p.detach().numpy()
        out[f"{name}_grad_gap"] = torch.autograd.grad((0.6 * -p[:, 1]).sum(), zt)[0].numpy()

        def tf_gap(_in, _nm, _em, _t):
            return -pred(_in, _nm, _em, _t)[:, 1]

        s = 400
        eps_n = rng_noise(2340 + ci, z.shape)
        s_t, t_t = torch.full((B, 1), s / T), torch.full((B, 1), (s + 1) / T)
        with InjectNoise([eps_n]), torch.no_grad():
            out[f"{name}_zs_unguided"] = model.sample_p_zs_given_zt(s_t, t_t, torch.from_numpy(z), tnm, tem, None).numpy()
        with InjectNoise([eps_n]):
            out[f"{name}_zs_guided"] = model.sample_p_zs_given_zt_guidance(s_t, t_t, torch.from_numpy(z), tnm, tem, tf_gap, 0.6).detach().numpy()
        out[f"{name}_z"], out[f"{name}_t"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"], out[f"{name}_step_noise"] = z, t, nm, em, eps_n
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, s=s, eseed=2300 + ci, pseed=2350 + ci, amp=amp,
                                                      over_e={k: v for k, v in over_e.items() if k != "diffusion_steps"}, over_p=over_p)))
    out["names"] = np.array(json.dumps([c[0] for c in cases]))
    save("g23_attention_tanh_flags", **out)


def g24_scalar_hyperparameters():
    """Every scalar hyper-parameter of the path away from its default at once (utils/args_edm.py, prediction_args.py):
    diffusion_noise_schedule 'polynomial_3' with diffusion_noise_precision 1e-4 (en_diffusion.py:47-61,191-218), normalize_factors
    [2, 3, 5] (:384-415), coords_range 7 (denoiser, egnn_new.py:290) and 4 (predictor, egnn_predictor/models.py:515), norm_constant 2,
    normalization_factor 2, inv_sublayers 2: gamma / coefficient tables, phi, predictor + gradient (amplified heads), and a guided
    T = 50 chain through sample_guidance on default-init weights, injected noise."""
    out = {}
    sched = dict(diffusion_noise_schedule="polynomial_3", diffusion_noise_precision=1e-4, normalize_factors=[2, 3, 5])
    for T in (50, 1000):
        sd = synth.synth_edm_state_dict(synth.edm_args(nf=8, n_layers=1), 1, seed=0)
        a, model = build_ref_edm("cata", sd, nf=8, n_layers=1, diffusion_steps=T, **sched)
        out[f"gamma_T{T}"] = model.gamma.gamma.numpy().copy()
        rows = []
        for s in ([0, 1, T // 2, T - 2, T - 1]):
            st = torch.full((1, 1), s) / T
            tt = (torch.full((1, 1), s) + 1) / T
            gs, gt = model.gamma(st), model.gamma(tt)
            zt = torch.zeros(1, 1, 1)
            s2, s_ts, a_ts = model.sigma_and_alpha_t_given_s(gt, gs, zt)
            sig_s, sig_t = model.sigma(gs, zt), model.sigma(gt, zt)
            rows.append([s, a_ts.item(), s2.item(), (s2 / a_ts / sig_t).item(), (s_ts * sig_s / sig_t).item(),
                         sig_s.item(), sig_t.item(), tt.item()])
        out[f"coef_T{T}"] = np.array(rows, dtype=np.float64)
    T = 50
    arch_e = dict(nf=32, n_layers=2, inv_sublayers=2, coords_range=7.0, norm_constant=2.0, normalization_factor=2.0)
    arch_p = dict(TINY_P, coords_range=4.0)
    for ci, (name, ds, nodes, mx) in enumerate([("cata", "cata", [4, 11, 7, 2, 11], 11), ("hetro", "hetro", [3, 5, 10, 7], 10)]):
        F = synth.num_node_features(ds)
        over = dict(diffusion_steps=T, **sched, **arch_e)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=2400 + ci, amplify_coord=True)
        a, model = build_ref_edm(ds, esd, **over)
        psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **arch_p), F, 5, seed=2410 + ci, amplify_coord=True)
        pa, pred = build_ref_pred(ds, psd, **arch_p)
        nm, em, z = case_inputs(ds, nodes, mx, seed=2420 + ci)
        B = z.shape[0]
        t = np.linspace(0.05, 0.95, B).astype(np.float32).reshape(B, 1)
        tnm, tem = torch.from_numpy(nm), torch.from_numpy(em)
        with torch.no_grad():
            out[f"{name}_eps"] = model.phi(torch.from_numpy(z), torch.from_numpy(t), tnm, tem, None).numpy()
        zt = torch.from_numpy(z).requires_grad_()
        p = pred(zt, tnm, tem, torch.from_numpy(t))
        out[f"{name}_pred"] = p.detach().numpy()
        out[f"{name}_grad_gap"] = torch.autograd.grad((0.6 * -p[:, 1]).sum(), zt)[0].numpy()
        out[f"{name}_z"], out[f"{name}_t"], out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = z, t, nm, em

        esd_c = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=2440 + ci)
        a_c, model_c = build_ref_edm(ds, esd_c, **over)
        psd_c = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **arch_p), F, 5, seed=2450 + ci)
        pa_c, pred_c = build_ref_pred(ds, psd_c, **arch_p)

        def tf_gap(_in, _nm, _em, _t):
            return -pred_c(_in, _nm, _em, _t)[:, 1]

        n = torch.tensor(nodes)
        Nn = max(nodes) * (2 if ds != "cata" else 1)
        noise = rng_noise(2430 + ci, (T + 2, len(nodes), Nn, 3 + F))
        with InjectNoise(list(noise)):
            x, h, nm2, em2 = ref_sampling.sample_guidance(a_c, model_c, tf_gap, n, scale=0.6, std=1.0)
        out[f"{name}_noise"], out[f"{name}_x_guided"], out[f"{name}_h_guided"] = noise, x.numpy(), h.numpy().astype(np.float32)
        out[f"{name}_chain_node_mask"], out[f"{name}_chain_edge_mask"] = nm2.numpy(), em2.numpy()
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, eseed=2400 + ci, pseed=2410 + ci, chain_eseed=2440 + ci,
                                                      chain_pseed=2450 + ci, nodes=nodes, over_e=dict(sched, **arch_e), over_p=arch_p)))
    save("g24_scalar_hyperparameters", **out)


def direct_z_target_torch(z, pred, nm):
    """-pred[:, 1] + 0.05 * sum over live nodes of |x_n|^2 + 0.02 * sum of the first feature column: depends on z through the
    predictor AND directly (numpy twin: tests/helpers.direct_z_target_grad)."""
    return -pred[:, 1] + 0.05 * ((z[:, :, :3] ** 2) * nm).sum((1, 2)) + 0.02 * (z[:, :, 3] * nm[:, :, 0]).sum(1)


def g21_direct_z_target():
    """sample_guidance with a closure that depends on z outside the predictor too (the reference differentiates any function
    of z_s, en_diffusion.py:899-903): tiny configs, T = 50, default-init weights, injected noise."""
    out = {}
    T = 50
    for ci, (name, ds, nodes) in enumerate([("cata", "cata", [6, 8, 8, 3]), ("hetro", "hetro", [3, 5, 4])]):
        F = synth.num_node_features(ds)
        over = dict(diffusion_steps=T, **TINY)
        esd = synth.synth_edm_state_dict(synth.edm_args(dataset=ds, **over), F, seed=2100 + ci)
        a, model = build_ref_edm(ds, esd, **over)
        psd = synth.synth_predictor_state_dict(synth.pred_args(dataset=ds, **TINY_P), F, 5, seed=2110 + ci)
        pa, pred = build_ref_pred(ds, psd, **TINY_P)

        def tf(_in, _nm, _em, _t):
            return direct_z_target_torch(_in, pred(_in, _nm, _em, _t), _nm)

        n = torch.tensor(nodes)
        Nn = max(nodes) * (2 if ds != "cata" else 1)
        noise = rng_noise(2120 + ci, (T + 2, len(nodes), Nn, 3 + F))
        with InjectNoise(list(noise)):
            x, h, nm, em = ref_sampling.sample_guidance(a, model, tf, n, scale=0.6, std=1.0)
        out[f"{name}_noise"], out[f"{name}_x"], out[f"{name}_h"] = noise, x.numpy(), h.numpy().astype(np.float32)
        out[f"{name}_node_mask"], out[f"{name}_edge_mask"] = nm.numpy(), em.numpy()
        out[f"{name}_cfg"] = np.array(json.dumps(dict(dataset=ds, T=T, eseed=2100 + ci, pseed=2110 + ci, nodes=nodes, amp=False)))
    save("g21_direct_z_target", **out)


def g8_checkpoint_roundtrip():
    """The reference's own loader must accept checkpoints written by gaudi_amd.synth.write_checkpoint
    (args.txt + model.pt, with and without the ``module.`` prefix).  Stores nothing but a marker."""
    import tempfile
    from utils.helpers import get_edm_args

    for dp in (True, False):
        with tempfile.TemporaryDirectory() as d:
            args = synth.edm_args(dp=dp, **TINY)
            sd = synth.synth_edm_state_dict(args, 1, seed=5)
            a0, m0 = build_ref_edm("cata", sd, **TINY)
            sd["gamma.gamma"] = m0.gamma.gamma.numpy().copy()
            synth.write_checkpoint(d, args, sd)
            a = get_edm_args(d)
            a.device = torch.device("cpu")
            model, _, _ = models_edm.get_model(a, FakeLoader(1, 5), only_norm=True)
            got = model.state_dict()
            pre = "module." if dp else ""
            for k, v in sd.items():
                assert np.array_equal(got[pre + k].numpy(), v), k
    print("g8: reference loader accepts synth checkpoints (dp=True/False)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22"]
    fns = dict(g1=g1_schedule, g2=g2_masks, g3=g3_phi, g4=g4_predictor, g5=g5_steps, g6=g6_decode,
               g7=g7_end_to_end, g8=g8_checkpoint_roundtrip, g9=g9_sample_chain, g10=g10_nonlinear_target, g11=g11_stability, g12=g12_ring_count_sampler, g13=g13_noised_predictor, g14=g14_long_chains, g15=g15_nan_scrub, g16=g16_fix_noise, g17=g17_nan_in_edge_gemm_matrix, g18=g18_large_molecules, g19=g19_amplified_default_steps, g20=g20_cosine_and_mean, g21=g21_direct_z_target, g22=g22_sin_embedding, g23=g23_attention_tanh_flags, g24=g24_scalar_hyperparameters)
    for w in which:
        fns[w]()
