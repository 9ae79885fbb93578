"""Debug helper: packed vs unpacked sampling calls, difference per molecule and per stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaudi_amd import synth
from gaudi_amd.engine import Engine
from gaudi_amd.sampling_edm import build_masks
dataset = sys.argv[1] if len(sys.argv) > 1 else "hetro"
rings = [3, 10, 4, 3, 5, 7, 3, 6] if dataset == "hetro" else [4, 11, 3, 5, 2, 6, 3, 4]
hetero = dataset == "hetro"
F = synth.num_node_features(dataset)
nm3, em_flat, N = build_masks(np.asarray(rings), max(rings), hetero)
B = len(rings)
nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
w = np.array([3, 0, 1, 1, 0] if hetero else [0, -1, 0, 0, 0], np.float32)
T = 9
eargs, pargs = synth.edm_args(nf=32, n_layers=2, diffusion_steps=T, dataset=dataset), synth.pred_args(nf=36, n_layers=3, dataset=dataset)
esd = synth.synth_edm_state_dict(eargs, F, seed=51, amplify_coord=False)
psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=52, amplify_coord=False)
rng = np.random.default_rng(8)
z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm[:, :, None]
z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / np.maximum(nm.sum(1)[:, None, None], 1) * nm[:, :, None]
eps = rng.standard_normal(z.shape).astype(np.float32)
res = {}
for pack in (1, 0):
    os.environ["GAUDI_PACK"] = str(pack)
    eng = Engine(0)
    eng.load_edm(eargs, esd); eng.load_predictor(pargs, psd)
    res[pack] = dict(step_u=eng.step(2, z, nm, em, eps), step_g=eng.step(2, z, nm, em, eps, target_w=w, scale=0.6),
                     dec=eng.decode(z, nm, em, eps)[0], samp_u=eng.sample(nm, em, seed=3, sample_offset=5)[0],
                     samp_g=eng.sample(nm, em, seed=3, sample_offset=5, target_w=w, scale=0.6)[0])
    eng.close()
for k in res[1]:
    a, b = res[1][k], res[0][k]
    d = np.abs(a - b).reshape(B, -1).max(1) / np.maximum(np.abs(b).reshape(B, -1).max(1), 1e-30)
    print(k, "rel diff per molecule:", " ".join(f"{v:.1e}" for v in d), flush=True)
