import os, sys
import numpy as np
sys.path.insert(0, ".")
from gaudi_amd import synth
from oracle import gaudi_oracle as O
from tests.helpers import rel_err
T = 6
eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
esd = synth.synth_edm_state_dict(eargs, 1, seed=31, amplify_coord=True)
psd = synth.synth_predictor_state_dict(pargs, 1, 5, seed=32, amplify_coord=True)
nm, em = O.build_masks([5, 11, 7, 3, 11], 11, False)
rng = np.random.default_rng(3)
z = O._combined_noise(rng.standard_normal((5, 11, 4)).astype(np.float32), nm)
t = np.full(5, 0.4, np.float32)
w = np.array([0, -1, 0, 0, 0], np.float32)
dp = np.broadcast_to(w * np.float32(0.6), (5, 5)).copy()
ref = O.edm_phi(esd, eargs, z, t, nm, em)
rp, rg = O.predictor_grad(psd, pargs, z, nm, em, t, dp)
from gaudi_amd.engine import Engine
for env in ({}, dict(GAUDI_WAVES="4"), dict(GAUDI_FORCE_GN="1"), dict(GAUDI_FORCE_GN8="1")):
    for k in ("GAUDI_WAVES", "GAUDI_FORCE_GN", "GAUDI_FORCE_GN8"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = Engine(0)
    eng.load_edm(eargs, esd); eng.load_predictor(pargs, psd)
    phi = eng.phi(z, t, nm, em)
    kv = eng.kernel_variant(), eng.node_buffers_global()
    pred, grad = eng.predictor_grad(z, t, nm, em, dp)
    print(env, kv, "phi", rel_err(phi, ref), "pred", rel_err(pred, rp), "grad", rel_err(grad, rg), flush=True)
    eng.close()
