"""Randomised cross-check of the two edge-GEMM arithmetic forms of the 8-wave kernels (and of the 4-wave family) on random
graphs and widths: the same guided / unguided reverse steps, denoiser and predictor-gradient calls must agree to fp32
rounding level.  Run on a GPU box:  python tools/fuzz_split_vs_fp32.py [cases]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gaudi_amd import synth
from gaudi_amd.engine import Engine
from gaudi_amd.sampling_edm import build_masks


def engine(env, eargs, esd, pargs, psd):
    saved = {k: os.environ.get(k) for k in ("GAUDI_EDGE_MATH", "GAUDI_WAVES")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        e = Engine(0)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    e.load_edm(eargs, esd)
    e.load_predictor(pargs, psd)
    return e


def rel(a, b):
    # max-norm difference; tensors that are small by cancellation (a one-node molecule's outputs) are measured against 0.05
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 0.05))


def main(cases, tol=1e-5):
    rng = np.random.default_rng(12345)
    worst = 0.0
    modes = {}
    for case in range(cases):
        hetero = bool(rng.integers(0, 2))
        ds = "hetro" if hetero else "cata"
        F = synth.num_node_features(ds)
        pairs = [(20, 36), (32, 36), (20, 20), (32, 32), (36, 36), (48, 40), (60, 60), (64, 64), (36, 48), (192, 196)]  # instantiated
        nf_e, nf_p = pairs[int(rng.integers(0, len(pairs) - 1))] if case % 7 != 6 else pairs[-1]
        Le, Lp = (2, 2) if nf_e < 192 else (3, 3)
        eargs = synth.edm_args(nf=nf_e, n_layers=Le, diffusion_steps=20, dataset=ds)
        pargs = synth.pred_args(nf=nf_p, n_layers=Lp, dataset=ds)
        esd = synth.synth_edm_state_dict(eargs, F, seed=case, amplify_coord=True)
        psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1000 + case, amplify_coord=True)
        B = int(rng.integers(1, 6))
        maxr = 10 if hetero else 11
        rings = [int(v) for v in rng.integers(1, maxr + 1, B)]
        nm, em, N = build_masks(rings, max(rings), hetero)
        if rng.integers(0, 3) == 0:  # knock out random edges (symmetric)
            e3 = em.reshape(B, N, N).copy()
            drop = rng.random((B, N, N)) < 0.3
            drop = drop | drop.transpose(0, 2, 1)
            e3[drop] = 0
            em = e3.reshape(em.shape)
        z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm
        z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1, keepdims=True), 1) * nm
        eps = rng.standard_normal(z.shape).astype(np.float32)
        w = rng.standard_normal(5).astype(np.float32)
        t = rng.uniform(0.05, 0.95, B).astype(np.float32)
        s = int(rng.integers(0, 20))
        outs = {}
        for name, env in (("split", {}), ("fp32", {"GAUDI_EDGE_MATH": "fp32"}), ("w4", {"GAUDI_WAVES": "4"})):
            e = engine(env, eargs, esd, pargs, psd)
            o = [e.phi(z, t, nm, em), e.step(s, z, nm, em, eps), e.step(s, z, nm, em, eps, target_w=w, scale=0.5)]
            if name == "split":
                modes[(e.kernel_variant()[1], e.edge_math()[1])] = modes.get((e.kernel_variant()[1], e.edge_math()[1]), 0) + 1
            o += list(e.predictor_grad(z, t, nm, em, np.broadcast_to(w, (B, 5)).copy()))
            outs[name] = o
            e.close()
        for name in ("fp32", "w4"):
            for k, (a, b) in enumerate(zip(outs["split"], outs[name])):
                r = rel(a, b)
                worst = max(worst, r)
                if not np.isfinite(a).all() or r > tol:
                    print(f"MISMATCH case {case} {ds} nf=({nf_e},{nf_p}) rings={rings} output {k} split vs {name}: {r:.3e} "
                          f"(max|a| {np.abs(a).max():.3e}, max|b| {np.abs(b).max():.3e}, max diff {np.abs(a - b).max():.3e})")
                    return 1
    print(f"{cases} random cases: split vs fp32-instruction vs 4-wave agree, worst relative difference {worst:.2e}; "
          f"(kernel family, edge math) of the default engine: {modes}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60))
