"""Debug helper: run the tiny / default GN-vs-LDS comparison step by step in this process (env decides the kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaudi_amd import synth
from gaudi_amd.engine import Engine
from oracle import gaudi_oracle as O
widths = sys.argv[1]
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
eng = Engine(0)
eng.load_edm(eargs, esd); eng.load_predictor(pargs, psd)
print("phi", flush=True); phi = eng.phi(z, t, nm, em); print(eng.kernel_variant(), float(np.abs(phi).max()), flush=True)
print("grad", flush=True); pred, grad = eng.predictor_grad(z, t, nm, em, np.broadcast_to(w * np.float32(0.6), (5, 5)).copy()); print(float(np.abs(grad).max()), flush=True)
print("sample unguided", flush=True); x, h, _ = eng.sample(nm, em, seed=4); print(float(np.abs(x).max()), flush=True)
print("step guided", flush=True); zs = eng.step(3, z, nm, em, rng.standard_normal(z.shape).astype(np.float32), target_w=w, scale=0.6); print(float(np.abs(zs).max()), flush=True)
print("sample guided", flush=True); x, h, _ = eng.sample(nm, em, seed=4, target_w=w, scale=0.6); print(float(np.abs(x).max()), flush=True)
print("OK", flush=True)
