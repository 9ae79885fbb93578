import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gaudi_oracle as O
from gaudi_amd import synth
from gaudi_amd.engine import Engine
from tests.helpers import nonlinear_target_grad
T = 6
F = synth.num_node_features("hetro")
eargs = synth.edm_args(dataset="hetro", diffusion_steps=T)
esd = synth.synth_edm_state_dict(eargs, F, seed=41)
pargs = synth.pred_args(dataset="hetro")
psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=42)
nodes = [5, 9, 10, 2]
nm, em = O.build_masks(nodes, 10, True)
B, N = nm.shape[0], nm.shape[1]
noise = np.random.default_rng(7).standard_normal((T + 2, B, N, 3 + F)).astype(np.float32)
eng = Engine(0); eng.load_edm(eargs, esd); eng.load_predictor(pargs, psd)
print("fused linear first", flush=True)
w = np.array([3,0,1,1,0], np.float32)
x, h, d = eng.sample(nm.reshape(B,N), em.reshape(B,N,N), noise=noise, target_w=w, scale=0.6)
print("ok fused", np.abs(x).max(), flush=True)
x, h, diag, z0 = eng.sample_callback(nm.reshape(B, N), em.reshape(B, N, N), nonlinear_target_grad, noise=noise, scale=0.6, return_z0=True)
print("ok cb", np.abs(x).max(), flush=True)
print("oracle...", flush=True)
xo, ho, zo = O.sample(esd, eargs, nm, em.reshape(B, N, N), noise, std=1.0, pred_sd=psd, pcfg=pargs, target_w=nonlinear_target_grad, scale=0.6)
print("ok oracle", np.abs(x - xo).max(), flush=True)
