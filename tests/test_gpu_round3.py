"""GPU evidence added in round 3: the RCCL code path executed on the hardware at hand (one rank), shard invariance of
heterogeneous batches through the plan hint, and the kernel-level changes of the round."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gaudi_amd import synth
from tests.helpers import edm_from_cfg, pred_from_cfg, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_gather_world1(tmp_path):
    """init_process_group("nccl", world_size=1, device_id=cuda:0) + gaudi_amd.dist.gather_to_all(..., device=dev) on the real
    engine's output + the plan check + bench.py's MAX all_reduce: the collective code of the N > 1 path runs over RCCL and
    returns the ungathered arrays bit for bit.  The RCCL version is logged (and kept in gpurun_out/ when that exists)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker_nccl.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=900)
    err = (tmp_path / "err0.txt").read_text() if (tmp_path / "err0.txt").exists() else ""
    assert r.returncode == 0, err + r.stderr[-2000:]
    info = json.loads((tmp_path / "nccl_world1.json").read_text())
    print("RCCL:", info)
    assert info["backend"] == "nccl" and info["same_x"] and info["same_h"] and info["finite"]
    assert (info["lo"], info["hi"]) == (0, 7) and info["allreduce"] == 1.25
    assert info["plan"][0] in (4, 8) and len(info["rccl_version"]) >= 2
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "rccl_world1.json"), "w") as f:
            json.dump(info, f)


def test_hetero_shards_run_one_plan_and_match_unsharded():
    """the kernel family / edge-GEMM arithmetic of a call follow from batch-wide maxima, so a shard without
    the batch's largest molecule could pick another plan.  With the plan hint (dist.sample_sharded(engine=...)) every shard
    runs the whole batch's plan and the concatenated shards equal the unsharded run bit for bit; without it they need not."""
    from gaudi_amd import dist as gdist
    from gaudi_amd.engine import Engine
    from gaudi_amd.sampling_edm import build_masks
    T = 12
    eargs = synth.edm_args(dataset="hetro", diffusion_steps=T)  # default widths: the LDS plan depends on the slot count
    pargs = synth.pred_args(dataset="hetro")
    F = synth.num_node_features("hetro")
    eng = Engine(0)
    eng.load_edm(eargs, synth.synth_edm_state_dict(eargs, F, seed=21))
    eng.load_predictor(pargs, synth.synth_predictor_state_dict(pargs, F, 5, seed=22))
    rings = np.array([3, 4, 3, 5, 4, 3, 10, 9])  # the big molecules sit in the second shard only
    nm3, em_flat, _ = build_masks(rings, 10, True)
    B, N = nm3.shape[0], nm3.shape[1]
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    w = np.array([3, 0, 1, 1, 0], np.float32)

    def sample_fn(nm_s, em_s, offset):
        x, h, _ = eng.sample(nm_s, em_s, seed=9, sample_offset=offset, target_w=w, scale=0.6)
        return x, h

    x_full, h_full = sample_fn(nm, em, 0)
    plan_full = (eng.kernel_variant()[1], eng.edge_math()[1])
    parts, plans = [], []
    for rank in range(2):
        lo, hi, x, h = gdist.sample_sharded(sample_fn, nm, em, rank, 2, engine=eng)
        parts.append((x, h))
        plans.append((eng.kernel_variant()[1], eng.edge_math()[1]))
    assert plans == [plan_full, plan_full], (plans, plan_full)
    assert np.array_equal(np.concatenate([p[0] for p in parts]), x_full)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), h_full)
    # the hint is cleared afterwards: a small batch on its own may plan differently again
    sample_fn(nm[:4], em[:4], 0)
    hint_free = (eng.kernel_variant()[1], eng.edge_math()[1])
    print("plans: whole batch", plan_full, "first shard alone", hint_free)
    eng.close()


# ------------------------------------------------------------------------------------------------ large molecules (N up to 40)
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


@pytest.mark.parametrize("widths", ["tiny", "default"])
def test_global_node_buffer_kernels_agree_with_the_lds_kernels(widths):
    """The V4G kernels (node buffers in a global scratch: the path molecules beyond the LDS limit take) run the SAME source
    as the 4-wave LDS kernels, only the address space of five buffers differs (hipcc contracts a few multiply-adds
    differently around global loads, so the last bit may differ): forced on a small batch they must agree to 2e-6 for phi,
    the predictor gradient and a guided chain -- and be reproducible run to run."""
    from oracle import gaudi_oracle as O
    T = 6
    if widths == "tiny":
        eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T), synth.pred_args(nf=36, n_layers=3)
    else:
        eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
    esd = synth.synth_edm_state_dict(eargs, 1, seed=31, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=32, amplify_coord=True)
    nm, em = O.build_masks([5, 11, 7, 3, 11], 11, False)
    rng = np.random.default_rng(3)
    z = O._combined_noise(rng.standard_normal((5, 11, 4)).astype(np.float32), nm)
    t = np.full(5, 0.4, np.float32)
    w = np.array([0, -1, 0, 0, 0], np.float32)
    res = []
    for env in (dict(GAUDI_WAVES=4), dict(GAUDI_FORCE_GN=1)):
        eng = _engine(eargs, esd, pargs, psd, **env)
        phi = eng.phi(z, t, nm, em)
        pred, grad = eng.predictor_grad(z, t, nm, em, np.broadcast_to(w * np.float32(0.6), (5, 5)).copy())
        x, h, _ = eng.sample(nm, em, seed=4, target_w=w, scale=0.6)
        assert eng.kernel_variant()[1] == 4
        res.append((phi, pred, grad, x, h))
        eng.close()
    for a, b in zip(*res):
        assert rel_err(a, b) < 2e-6
    eng = _engine(eargs, esd, pargs, psd, GAUDI_FORCE_GN=1)
    x2, h2, _ = eng.sample(nm, em, seed=4, target_w=w, scale=0.6)
    eng.close()
    assert np.array_equal(x2, res[1][3]) and np.array_equal(h2, res[1][4])


@pytest.mark.parametrize("N", [24, 32, 40])
def test_large_molecules_vs_oracle(N):
    """Complete graphs of 24 / 32 / 40 nodes (and smaller molecules padded to the same N) at the DEFAULT widths: phi, the
    predictor gradient and a guided step against the oracle at 1e-4.  These do not fit 160 KiB of LDS; the call must pick
    the global-node-buffer kernels by itself."""
    from oracle import gaudi_oracle as O
    T = 1000
    eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
    esd = synth.synth_edm_state_dict(eargs, 1, seed=41, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=42, amplify_coord=True)
    nm, em = O.build_masks([N, N - 7, 5], N, False)
    rng = np.random.default_rng(N)
    z = O._combined_noise(rng.standard_normal((3, N, 4)).astype(np.float32), nm)
    eps = rng.standard_normal((3, N, 4)).astype(np.float32)
    s = 300
    t = np.full(3, np.float32(s + 1) / np.float32(T), np.float32)
    w = O.target_max_gap_weights(5)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    eng = _engine(eargs, esd, pargs, psd)
    assert rel_err(eng.phi(z, t, nm, em), O.edm_phi(esd, eargs, z, t, nm, em)) < 1e-4  # (N = 24 alone still fits an 8-wave EDM kernel)
    dp = np.broadcast_to(w * np.float32(0.6), (3, 5)).copy()
    pred, grad = eng.predictor_grad(z, t, nm, em, dp)
    # the predictor of a graph this size keeps its node buffers in global memory: on the 8-wave kernels (V8G, round 4) while no
    # node has more than 32 live edges (complete graphs up to 33 nodes), on the 4-wave V4G kernels beyond
    assert eng.node_buffers_global() and eng.kernel_variant()[1] == (8 if N <= 33 else 4)
    opred, ograd = O.predictor_grad(psd, pargs, z, nm, em, t, dp)
    assert rel_err(pred, opred) < 1e-4 and rel_err(grad, ograd) < 1e-4
    assert np.abs(grad * (1 - nm)).max() == 0
    zs = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
    assert rel_err(zs, O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6)) < 1e-4
    x, h, d = eng.sample(nm, em, seed=1, target_w=w, scale=0.6)  # a whole (short) chain runs too
    eng.close()


def test_n40_hetero_vs_reference(golden):
    """Hetero 20 rings = 40 graph nodes, default architectures, against the REFERENCE's own outputs (g18): phi, predictor +
    input gradient, unguided and guided teacher-forced step at 1e-4."""
    import json as _json
    g = golden("g18_large_molecules")
    cfg = _json.loads(str(g["cfg"]))
    T, s = cfg["T"], cfg["s"]
    base = dict(dataset=cfg["dataset"], amp=True)
    eargs, esd = edm_from_cfg(dict(base, wseed=cfg["eseed"]), diffusion_steps=T)
    pargs, psd = pred_from_cfg(dict(base, wseed=cfg["pseed"]))
    z, nm, em, eps = g["z"], g["node_mask"], g["edge_mask"], g["eps"]
    t = np.full(z.shape[0], np.float32(s + 1) / np.float32(T), np.float32)
    w = np.array([0, -1, 0, 0, 0], np.float32)
    # the default (round 4): the V8G kernels -- 8 waves, node buffers in global memory; GAUDI_GN8=0: round 3's V4G kernels (4 waves)
    for env, waves in (({}, 8), ({"GAUDI_GN8": 0}, 4)):
        eng = _engine(eargs, esd, pargs, psd, **env)
        assert rel_err(eng.phi(z, t, nm, em), g["phi"]) < 1e-4
        pred, grad = eng.predictor_grad(z, t, nm, em, np.broadcast_to(w * np.float32(0.6), (z.shape[0], 5)).copy())
        assert rel_err(pred, g["pred"]) < 1e-4 and rel_err(grad, g["grad_gap"]) < 1e-4
        assert rel_err(eng.step(s, z, nm, em, eps), g["zs_unguided"]) < 1e-4
        assert rel_err(eng.step(s, z, nm, em, eps, target_w=w, scale=0.6), g["zs_guided"]) < 1e-4
        assert eng.kernel_variant()[1] == waves and eng.node_buffers_global()
        eng.close()


# ------------------------------------------------------------------------------------------------ reference-held anchor (g19)
@pytest.mark.parametrize("name", ["cata", "hetro"])
@pytest.mark.parametrize("math", ["split", "fp32"])
def test_amplified_default_architecture_steps_vs_reference(golden, name, math):
    """VERDICT r2 weak #1b: the amplified-head default-architecture step used to be held only against the builder's own
    restatement.  g19 = the REFERENCE's teacher-forced unguided / guided steps at s = 999, 400, 0 in fp32 and in float64
    (C3 and C4 shapes): the default kernels (split-operand edge GEMMs) and the fp32-instruction kernels sit within 1e-4 of
    both."""
    import json as _json
    g = golden("g19_amplified_default_steps")
    cfg = _json.loads(str(g[name + "_cfg"]))
    T = cfg["T"]
    base = dict(dataset=cfg["dataset"], amp=True)
    eargs, esd = edm_from_cfg(dict(base, wseed=cfg["eseed"]), diffusion_steps=T)
    pargs, psd = pred_from_cfg(dict(base, wseed=cfg["pseed"]))
    eng = _engine(eargs, esd, pargs, psd, **({} if math == "split" else dict(GAUDI_EDGE_MATH="fp32")))
    z, nm, em, w = g[name + "_z"], g[name + "_node_mask"], g[name + "_edge_mask"], g[name + "_w"]
    for s in (999, 400, 0):
        eps = g[f"{name}_s{s}_eps"]
        zu, zg = eng.step(s, z, nm, em, eps), eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
        assert eng.kernel_variant()[1] == 8 and (eng.edge_math()[1] != 0) == (math == "split")
        for tag in ("fp32", "fp64"):
            assert rel_err(zu, g[f"{name}_s{s}_zs_unguided_{tag}"]) < 1e-4, (s, tag)
            assert rel_err(zg, g[f"{name}_s{s}_zs_guided_{tag}"]) < 1e-4, (s, tag)
    eng.close()


# ------------------------------------------------------------------------------------------------ the reference's closure targets
def test_reference_style_closure_targets_and_call_shape(golden):
    """generation_guidance.py:190-211 as written: get_model(...), then get_cond_predictor_model(args, dataset) with NO model
    argument, then a target function that is a plain closure over (z, node_mask, edge_mask, t) calling cond_predictor --
    the linear gap target and a nonlinear one (g10: the reference's own guided chain with that nonlinear closure)."""
    import json as _json
    import types

    import torch

    from gaudi_amd import sampling_edm
    from gaudi_amd._lib import GaudiError
    from gaudi_amd.models_edm import get_cond_predictor_model, get_model, target_function_max_gap
    from tests.helpers import TINY, TINY_P
    g = golden("g10_nonlinear_target")
    cfg = _json.loads(str(g["chain_cfg"]))
    eargs, esd = edm_from_cfg(dict(dataset=cfg["dataset"], over=TINY, wseed=cfg["eseed"], amp=False), diffusion_steps=cfg["T"])
    pargs, psd = pred_from_cfg(dict(dataset=cfg["dataset"], over=TINY_P, wseed=cfg["pseed"], amp=False))
    model, _, _ = get_model(eargs, state_dict=esd)
    cond_predictor = get_cond_predictor_model(pargs, None, state_dict=psd)  # (args, dataset): attaches to the model made last
    assert cond_predictor.engine is model.engine
    model.injected_noise = g["chain_noise"]
    args = types.SimpleNamespace(device="cuda", dataset="hetro", max_nodes=10)

    def target_nonlinear(_input, _node_mask, _edge_mask, _t):  # the closure tools/make_golden.py gave the reference
        p = cond_predictor(_input, _node_mask, _edge_mask, _t)
        return 0.5 * torch.log1p(p[:, 1] ** 2) + 0.1 * torch.tanh(p[:, 0]) * p[:, 3] + _t[:, 0] * p[:, 2]

    x, h, nm, em = sampling_edm.sample_guidance(args, model, target_nonlinear, cfg["nodes"], scale=0.6)
    assert rel_err(x.numpy(), g["chain_x"]) < 1e-4 and np.array_equal(h.numpy(), g["chain_h"])

    def target_function_max_gap_ref(_input, _node_mask, _edge_mask, _t):  # generation_guidance.py:200-203 verbatim shape
        return -cond_predictor(_input, _node_mask, _edge_mask, _t)[:, 1]

    xa, ha, _, _ = sampling_edm.sample_guidance(args, model, target_function_max_gap_ref, cfg["nodes"], scale=0.6)
    xb, hb, _, _ = sampling_edm.sample_guidance(args, model, target_function_max_gap(cond_predictor), cfg["nodes"], scale=0.6)
    assert rel_err(xa.numpy(), xb.numpy()) < 1e-4 and np.array_equal(ha.numpy(), hb.numpy())  # callback path vs fused path

    def target_direct(_input, _node_mask, _edge_mask, _t):  # depends on z outside the predictor too: accepted since round 4
        return 0.01 * ((_input[:, :, :3] ** 2) * _node_mask).sum((1, 2)) - cond_predictor(_input, _node_mask, _edge_mask, _t)[:, 1]

    xd, hd, _, _ = sampling_edm.sample_guidance(args, model, target_direct, cfg["nodes"], scale=0.6)  # (vs the reference: g21)
    assert np.isfinite(xd.numpy()).all() and rel_err(xd.numpy(), xa.numpy()) > 1e-4
    model.engine.close()


# ------------------------------------------------------------------------------------------------ packed workgroups
@pytest.mark.parametrize("dataset,rings", [("hetro", [3, 10, 4, 3, 5, 7, 3, 6, 4, 9, 3, 4, 8, 5, 3, 3]),
                                           ("cata", [4, 11, 3, 5, 2, 6, 3, 4, 1, 7, 5, 2])])
def test_packed_workgroups_equal_unpacked_bit_for_bit(dataset, rings):
    """Sampling calls pack small molecules into one workgroup as components of a disjoint graph (gaudi_hip.hip:
    pack_groups).  A molecule keeps its own tiles, node order, noise stream and per-molecule reductions, so guided and
    unguided chains, injected noise, fix_noise and teacher-forced steps give the SAME BITS with GAUDI_PACK=0 -- at the
    default widths (where the LDS plan decides) and at the test widths."""
    from gaudi_amd.sampling_edm import build_masks
    hetero = dataset == "hetro"
    F = synth.num_node_features(dataset)
    nm3, em_flat, N = build_masks(np.asarray(rings), max(rings), hetero)
    B = len(rings)
    nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    w = np.array([3, 0, 1, 1, 0] if hetero else [0, -1, 0, 0, 0], np.float32)
    for widths in ("tiny", "default"):
        T = 9 if widths == "tiny" else 5
        if widths == "tiny":
            eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T, dataset=dataset), synth.pred_args(nf=36, n_layers=3, dataset=dataset)
        else:
            eargs, pargs = synth.edm_args(diffusion_steps=T, dataset=dataset), synth.pred_args(dataset=dataset)
        esd = synth.synth_edm_state_dict(eargs, F, seed=51, amplify_coord=True)
        psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=52, amplify_coord=True)
        rng = np.random.default_rng(8)
        noise = rng.standard_normal((T + 2, B, N, 3 + F)).astype(np.float32)
        z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm[:, :, None]
        z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1)[:, None, None], 1) * nm[:, :, None]
        eps = rng.standard_normal(z.shape).astype(np.float32)
        outs = []
        for pack in (1, 0):
            eng = _engine(eargs, esd, pargs, psd, GAUDI_PACK=pack)
            o = [eng.sample(nm, em, seed=3, sample_offset=5, target_w=w, scale=0.6, return_z0=True),
                 eng.sample(nm, em, seed=3, sample_offset=5, return_z0=True),
                 eng.sample(nm, em, noise=noise, target_w=w, scale=0.6, return_z0=True)]
            assert eng.kernel_variant()[1] == 8
            o.append((eng.step(T - 2, z, nm, em, eps, target_w=w, scale=0.6), eng.step(2, z, nm, em, eps), None))
            eng.set_fix_noise(True)
            o.append(eng.sample(nm, em, seed=3, sample_offset=5, target_w=w, scale=0.6, return_z0=True))
            eng.close()
            outs.append(o)
        for a, b in zip(*outs):
            for u, v in zip(a, b):
                if isinstance(u, np.ndarray):
                    assert np.array_equal(u, v), (widths, dataset)
                elif isinstance(u, dict):
                    assert u["nan_count"] == v["nan_count"] and u["max_masked_leak"] == v["max_masked_leak"] == 0
        x = outs[0][0][0]
        assert np.isfinite(x).all() and len({x[i].tobytes() for i in range(B)}) == B  # every molecule its own noise
        if widths == "tiny":
            # NaN scrubbing is per molecule = per component: a NaN planted in the last coordinate head (every molecule's
            # velocity becomes NaN and is scrubbed) and one in the predictor readout (z_s scrubbed to zeros)
            res = []
            for pack in (1, 0):
                esd_bad = {k: v.copy() for k, v in esd.items()}
                esd_bad[f"dynamics.egnn.e_block_{eargs['n_layers'] - 1}.gcl_equiv.coord_mlp.4.weight"][0, 3] = np.nan
                psd_bad = {k: v.copy() for k, v in psd.items()}
                psd_bad["egnn.embedding_out.weight"][1, 5] = np.nan
                e1, e2 = _engine(eargs, esd_bad, pargs, psd, GAUDI_PACK=pack), _engine(eargs, esd, pargs, psd_bad, GAUDI_PACK=pack)
                a = e1.sample(nm, em, seed=3, target_w=w, scale=0.6)
                b = e2.sample(nm, em, seed=3, target_w=w, scale=0.6)
                c = e1.step(T - 2, z, nm, em, eps)
                res.append((a[0], a[1], b[0], b[1], c, a[2]["nan_count"], b[2]["nan_count"]))
                e1.close(); e2.close()
            live = nm != 0
            for i, (u, v) in enumerate(zip(*res)):
                if isinstance(u, np.ndarray):
                    assert np.array_equal(u[live], v[live]), i      # every live node: the same bits
                    assert np.all(u[~live] == 0), i                  # masked nodes have no slot in a packed workgroup: zeros
                elif i == 6:
                    assert u > 0 and v > 0, i  # poisoned predictor: z_s scrubbed (the count itself includes masked rows when unpacked)
            assert np.isfinite(res[0][0]).all() and np.isfinite(res[0][4]).all()


def test_packed_workgroups_fuzz():
    """Random hetero batches (ring counts 1..10, 2..40 molecules, test widths): the packed launch, the unpacked launch and
    every molecule sampled ALONE (same padded N, its own sample index) give the same bits -- a molecule's result depends on
    nothing but its own graph, its sample index and the weights."""
    from gaudi_amd.sampling_edm import build_masks
    dataset, F, T = "hetro", synth.num_node_features("hetro"), 6
    eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T, dataset=dataset), synth.pred_args(nf=36, n_layers=3, dataset=dataset)
    esd = synth.synth_edm_state_dict(eargs, F, seed=61, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=62, amplify_coord=True)
    w = np.array([1, -2, 0.5, 0, 1], np.float32)
    packed, plain = _engine(eargs, esd, pargs, psd, GAUDI_PACK=1), _engine(eargs, esd, pargs, psd, GAUDI_PACK=0)
    rng = np.random.default_rng(2024)
    groups_seen = 0
    for case in range(12):
        B = int(rng.integers(2, 41))
        rings = rng.integers(1, 11, size=B)
        nm3, em_flat, N = build_masks(rings, int(rings.max()), True)
        nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
        off = int(rng.integers(0, 1000))
        guided = case % 3 != 2
        kw = dict(target_w=w, scale=0.7) if guided else {}
        a = packed.sample(nm, em, seed=17, sample_offset=off, return_z0=True, **kw)
        b = plain.sample(nm, em, seed=17, sample_offset=off, return_z0=True, **kw)
        assert packed.kernel_variant()[1] == 8
        for u, v in zip(a, b):
            if isinstance(u, np.ndarray):
                assert np.array_equal(u, v), case
        assert np.isfinite(a[0]).all()
        G = packed.pack_plan(nm, em)[0]
        groups_seen += int(G < B)
        for i in rng.choice(B, size=min(3, B), replace=False):
            alone = plain.sample(nm[i:i + 1], em[i:i + 1], seed=17, sample_offset=off + int(i), return_z0=True, **kw)
            assert np.array_equal(alone[0][0], a[0][i]) and np.array_equal(alone[1][0], a[1][i]), (case, int(i))
    assert groups_seen >= 10  # the batches really were packed
    packed.close(); plain.close()


def test_default_fused_kernel_200_repeats_are_bit_identical():
    """The LDS-DMA weight ring's loads are not tracked by the compiler's barrier fence (every wave retires its
    own with vmcnt(0) before the trip barrier).  A lost wait shows as run-to-run differences under load: 200 guided
    launches of the default fused kernel (sampler_kernel_v<V8S,192,208>, C3 shape: 11-node cata molecules, one per CU on
    all 256 CUs, 3 reverse steps + decode each) must return the same bits."""
    F, N, B, T = 1, 11, 256, 3
    eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
    esd = synth.synth_edm_state_dict(eargs, F, seed=81, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=82, amplify_coord=True)
    eng = _engine(eargs, esd, pargs, psd)
    nm = np.ones((B, N), np.float32)
    em = np.broadcast_to(1.0 - np.eye(N, dtype=np.float32), (B, N, N)).copy()
    w = np.array([0, -1, 0, 0, 0], np.float32)
    ref = eng.sample(nm, em, seed=5, target_w=w, scale=0.6, return_z0=True)
    assert eng.kernel_variant()[1] == 8 and eng.edge_math()[1] == 1 and np.isfinite(ref[0]).all()
    for rep in range(200):
        out = eng.sample(nm, em, seed=5, target_w=w, scale=0.6, return_z0=True)
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]) and np.array_equal(out[-1], ref[-1]), rep
    eng.close()


# ------------------------------------------------------------------------------------------------ more than one round of edge tiles
@pytest.mark.parametrize("widths", ["tiny", "default"])
@pytest.mark.parametrize("N", [12, 14, 16, 19])
def test_dense_molecules_beyond_128_edge_slots_stay_on_8_waves(widths, N):
    """A fully connected molecule of 12+ nodes has more than 128 live edges = more than one round of eight 16-slot tiles.
    The guided path used to drop such calls to the 4-wave kernels; the 8-wave predictor now runs the
    rounds one after the other (the reverse pass parks du of every tile in the stash between its chains and the publish
    phase).  Predictor + input gradient and a guided step against the oracle at 1e-4, agreement with the 4-wave kernels
    (GAUDI_PRED_ROUNDS=0), bitwise repeatability, and a mixed batch (one big, two small molecules) through a whole chain."""
    from oracle import gaudi_oracle as O
    T = 1000 if widths == "default" else 12
    if widths == "tiny":
        eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T), synth.pred_args(nf=36, n_layers=3)
    else:
        eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
    esd = synth.synth_edm_state_dict(eargs, 1, seed=91, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=92, amplify_coord=True)
    nm, em = O.build_masks([N, N - 3, 5], N, False)
    em = np.asarray(em, np.float32).reshape(3, N, N)
    rng = np.random.default_rng(100 + N)
    z = O._combined_noise(rng.standard_normal((3, N, 4)).astype(np.float32), nm)
    eps = rng.standard_normal((3, N, 4)).astype(np.float32)
    s = T // 3
    t = np.full(3, np.float32(s + 1) / np.float32(T), np.float32)
    w = O.target_max_gap_weights(5)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    dp = np.broadcast_to(w * np.float32(0.6), (3, 5)).copy()
    eng = _engine(eargs, esd, pargs, psd)
    pred, grad = eng.predictor_grad(z, t, nm, em, dp)
    assert eng.kernel_variant()[1] == 8, "the 8-wave predictor must take a graph of more than 128 edge slots"
    opred, ograd = O.predictor_grad(psd, pargs, z, nm, em, t, dp)
    assert rel_err(pred, opred) < 1e-4 and rel_err(grad, ograd) < 1e-4
    assert np.abs(grad * (1 - nm)).max() == 0
    zs = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
    assert eng.kernel_variant()[1] == 8
    assert rel_err(zs, O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6)) < 1e-4
    for _ in range(3):
        assert np.array_equal(eng.step(s, z, nm, em, eps, target_w=w, scale=0.6), zs)
    one = eng.step(s, z[:1], nm[:1], em[:1], eps[:1], target_w=w, scale=0.6)  # the big molecule alone: the same bits
    assert np.array_equal(one[0], zs[0])
    x, h, d = eng.sample(nm, em, seed=1, target_w=w, scale=0.6) if widths == "tiny" else (None, None, None)
    eng.close()
    old = _engine(eargs, esd, pargs, psd, GAUDI_PRED_ROUNDS=0)
    pred4, grad4 = old.predictor_grad(z, t, nm, em, dp)
    assert old.kernel_variant()[1] == 4
    assert rel_err(pred4, pred) < 1e-5 and rel_err(grad4, grad) < 2e-5
    if widths == "tiny":
        x4, h4, _ = old.sample(nm, em, seed=1, target_w=w, scale=0.6)
        assert np.isfinite(x).all() and rel_err(x, x4) < 1e-3 and np.array_equal(h, h4)
    old.close()


def test_mixed_batch_big_and_small_molecules_packs_and_matches_unpacked():
    """One fully connected 13-node molecule (10 edge tiles: two rounds, MR kernels) among small hetero-like ones: the small
    ones still share workgroups, and the result equals the unpacked run bit for bit."""
    from oracle import gaudi_oracle as O
    T = 6
    eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T), synth.pred_args(nf=36, n_layers=3)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=71, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=72, amplify_coord=True)
    N = 13
    sizes = [13, 3, 4, 2, 5, 3, 13, 4]
    nm, em = O.build_masks(sizes, N, False)
    em = np.asarray(em, np.float32).reshape(len(sizes), N, N)
    w = np.array([0, -1, 0.5, 0, 0], np.float32)
    outs = []
    for pack in (1, 0):
        eng = _engine(eargs, esd, pargs, psd, GAUDI_PACK=pack)
        outs.append(eng.sample(nm, em, seed=9, sample_offset=3, target_w=w, scale=0.6, return_z0=True))
        assert eng.kernel_variant()[1] == 8
        if pack:
            assert eng.pack_plan(nm, em)[0] < len(sizes)
        eng.close()
    for u, v in zip(*outs):
        if isinstance(u, np.ndarray):
            assert np.array_equal(u, v)
    assert np.isfinite(outs[0][0]).all()


@pytest.mark.parametrize("shape", ["c2", "c3", "c4"])
def test_full_batch_guided_step_vs_cpp_port(shape):
    """BASELINE's full batch shapes through ONE teacher-forced step at the default architectures against the C++/OpenMP
    restatement (oracle/gaudi_cpu.cpp, itself pinned to the reference's goldens): C2 = 256 cata molecules of 11 nodes, UNGUIDED
    (VERDICT r4 weak #1b: the unguided full batch was never compared molecule by molecule), C3 = the same batch guided (one
    workgroup per CU), C4 = 1024 hetero molecules of 3-10 rings, packed into ~700 workgroups.  Every molecule of the batch is
    compared -- block -> molecule order, packing maps and noise rows included -- at 1e-4."""
    from oracle import build_cpu
    from oracle import gaudi_oracle as O
    if not build_cpu.cpu_ok():
        pytest.skip("host CPU lacks the ISA the C++ port is built for")
    from gaudi_amd.sampling_edm import build_masks
    T = 1000
    hetero = shape == "c4"
    ds = "hetro" if hetero else "cata"
    F = synth.num_node_features(ds)
    eargs, pargs = synth.edm_args(diffusion_steps=T, dataset=ds), synth.pred_args(dataset=ds)
    esd = synth.synth_edm_state_dict(eargs, F, seed=0, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1, amplify_coord=True)
    if hetero:
        B = 1024
        rings = np.random.default_rng(1).integers(3, 11, size=B)
        nm3, em_flat, N = build_masks(rings, 10, True)
        nm, em = nm3.reshape(B, N).astype(np.float32), em_flat.reshape(B, N, N).astype(np.float32)
        w = np.array([3, 0, 1, 1, 0], np.float32)
    else:
        B, N = 256, 11
        nm = np.ones((B, N), np.float32)
        em = np.broadcast_to(1.0 - np.eye(N, dtype=np.float32), (B, N, N)).copy()
        w = np.array([0, -1, 0, 0, 0], np.float32)
    rng = np.random.default_rng(77)
    z = O._combined_noise(rng.standard_normal((B, N, 3 + F)).astype(np.float32), nm[:, :, None])
    eps = rng.standard_normal((B, N, 3 + F)).astype(np.float32)
    s = 500
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    if shape == "c2":
        w = None
    port = build_cpu.CpuPort()
    port.load_edm(eargs, esd)
    port.load_predictor(pargs, psd)
    want = port.step(O.step_coefficients(gamma, s, s + 1), np.float32(np.float32(s + 1) / np.float32(T)), z, nm, em, eps,
                     target_w=w, scale=0.6)
    port.close()
    eng = _engine(eargs, esd, pargs, psd)
    got = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
    assert eng.kernel_variant()[1] == 8
    eng.close()
    per_mol = np.abs(got - want).reshape(B, -1).max(1) / np.abs(want).reshape(B, -1).max(1)
    assert per_mol.max() < 1e-4, (int(per_mol.argmax()), float(per_mol.max()))
    assert np.all(got[nm == 0] == 0)


def test_random_masks_fuzz_vs_oracle():
    """Random ragged batches through a guided and an unguided step against the oracle: random live-node counts (including
    empty molecules and single nodes), random symmetric and ASYMMETRIC edge masks, isolated live nodes, live edges between
    masked nodes (dropped, as in the reference where they reach nothing), fractional mask values excluded.  Packed launches
    on (the default): 30 cases, 1e-4 per molecule."""
    from oracle import gaudi_oracle as O
    T = 10
    F = 3
    eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T), synth.pred_args(nf=36, n_layers=2)
    esd = synth.synth_edm_state_dict(eargs, F, seed=31, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=32, amplify_coord=True)
    gamma = O.gamma_table("polynomial_2", T, 1e-5)
    w = np.array([0.5, -1.0, 0.25, 0.0, 1.0], np.float32)
    eng = _engine(eargs, esd, pargs, psd)
    rng = np.random.default_rng(4242)
    worst = 0.0
    for case in range(30):
        B, N = int(rng.integers(1, 12)), int(rng.integers(1, 13))
        n_live = rng.integers(0, N + 1, size=B)
        if case % 5 == 0:
            n_live[rng.integers(0, B)] = 0          # an empty molecule
        if case % 7 == 0:
            n_live[rng.integers(0, B)] = 1          # a single node
        nm = (np.arange(N)[None, :] < n_live[:, None]).astype(np.float32)
        dens = rng.uniform(0.15, 1.0)
        em = (rng.random((B, N, N)) < dens).astype(np.float32) * (1 - np.eye(N, dtype=np.float32))[None]
        if case % 2 == 0:
            em = np.maximum(em, em.transpose(0, 2, 1))  # symmetric; odd cases keep directed edges
        if case % 3:
            em *= nm[:, :, None] * nm[:, None, :]      # otherwise some edges touch masked nodes
        z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm[:, :, None]
        cnt = np.maximum(nm.sum(1), 1)[:, None, None]
        z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / cnt * nm[:, :, None]
        eps = rng.standard_normal(z.shape).astype(np.float32)
        s = int(rng.integers(0, T))
        for guided in (True, False):
            got = eng.step(s, z, nm, em, eps, target_w=w if guided else None, scale=0.7)
            want = (O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm[:, :, None], em, eps, w, 0.7) if guided
                    else O.step_unguided(esd, eargs, gamma, s, z, nm[:, :, None], em, eps))
            assert np.isfinite(got).all(), (case, guided)
            assert np.all(got[nm == 0] == 0), (case, guided)
            den = np.maximum(np.abs(want).reshape(B, -1).max(1), 1e-3)
            err = float((np.abs(got - want).reshape(B, -1).max(1) / den).max())
            worst = max(worst, err)
            assert err < 1e-4, (case, guided, B, N, n_live.tolist(), err)
    eng.close()


def test_callback_target_on_a_graph_of_several_rounds():
    """An arbitrary (nonlinear) target through gaudi_sample_cb on a fully connected 13-node molecule: the two device phases
    per step (predictor forward | reverse pass + update) run on the MR 8-wave kernels; with a LINEAR target given as a
    callback the chain equals the fused linear-target chain bit for bit, and a nonlinear one agrees with the 4-wave kernels."""
    from oracle import gaudi_oracle as O
    T = 8
    eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T), synth.pred_args(nf=36, n_layers=3)
    esd = synth.synth_edm_state_dict(eargs, 1, seed=61, amplify_coord=True)
    psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=62, amplify_coord=True)
    N = 13
    nm, em = O.build_masks([13, 6, 13], N, False)
    em = np.asarray(em, np.float32).reshape(3, N, N)
    w = np.array([0.5, -1.0, 0.25, 0.0, 1.0], np.float32)
    lin = lambda pred, t: np.broadcast_to(w, pred.shape)
    nonlin = lambda pred, t: (2.0 * pred * w + np.cos(pred[:, :1]) * 0.1).astype(np.float32)
    eng = _engine(eargs, esd, pargs, psd)
    a = eng.sample(nm, em, seed=4, target_w=w, scale=0.5)
    b = eng.sample_callback(nm, em, lin, seed=4, scale=0.5)
    assert eng.kernel_variant()[1] == 8
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c = eng.sample_callback(nm, em, nonlin, seed=4, scale=0.5)
    eng.close()
    old = _engine(eargs, esd, pargs, psd, GAUDI_PRED_ROUNDS=0)
    c4 = old.sample_callback(nm, em, nonlin, seed=4, scale=0.5)
    assert old.kernel_variant()[1] == 4
    old.close()
    assert np.isfinite(c[0]).all() and rel_err(c[0], c4[0]) < 1e-3 and np.array_equal(c[1], c4[1])


def test_two_handles_sample_concurrently_from_two_threads():
    """The C ABI is re-entrant across handles (include/gaudi_hip.h): two engines with DIFFERENT weights and widths load and
    sample at the same time from two host threads (ctypes drops the GIL in the calls; each handle has its own stream) and
    return exactly what they return one after the other."""
    import threading
    from gaudi_amd.sampling_edm import build_masks
    cfgs = []
    for k, (nf_e, nf_p, ds) in enumerate(((32, 36, "hetro"), (64, 60, "cata"))):
        F = synth.num_node_features(ds)
        eargs = synth.edm_args(nf=nf_e, n_layers=2, diffusion_steps=12, dataset=ds)
        pargs = synth.pred_args(nf=nf_p, n_layers=2, dataset=ds)
        esd = synth.synth_edm_state_dict(eargs, F, seed=10 + k, amplify_coord=True)
        psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=20 + k, amplify_coord=True)
        rings = np.random.default_rng(k).integers(2, 9, size=40)
        nm3, em_flat, N = build_masks(rings, int(rings.max()), ds == "hetro")
        cfgs.append((eargs, esd, pargs, psd, nm3.reshape(40, N), em_flat.reshape(40, N, N)))
    w = np.array([0.5, -1.0, 0.25, 0.0, 1.0], np.float32)

    def run(cfg, out, reps):
        eargs, esd, pargs, psd, nm, em = cfg
        for r in range(reps):
            eng = _engine(eargs, esd, pargs, psd)  # loading (= weight packing) inside the thread too
            out.append(eng.sample(nm, em, seed=3 + r, target_w=w, scale=0.6))
            eng.close()

    serial = [[], []]
    for k in range(2):
        run(cfgs[k], serial[k], 3)
    both = [[], []]
    threads = [threading.Thread(target=run, args=(cfgs[k], both[k], 3)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(2):
        assert len(both[k]) == 3
        for a, b in zip(serial[k], both[k]):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
