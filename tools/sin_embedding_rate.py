"""Rate of a sin_embedding denoiser (4-wave kernels, kernse_*.hip) next to the ordinary one (8-wave kernels) on the C3 shape:
cata 11 rings, default widths, guided, B molecules, T steps.  python tools/sin_embedding_rate.py [B] [T]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gaudi_amd import synth
from gaudi_amd.engine import Engine
from gaudi_amd.sampling_edm import build_masks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
F = synth.num_node_features("cata")
nm3, em_flat, N = build_masks([11] * B, 11, False)
nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
w = np.array([0, -1, 0, 0, 0], np.float32)
for se in (False, True):
    eargs, pargs = synth.edm_args(dataset="cata", diffusion_steps=T, sin_embedding=se), synth.pred_args(dataset="cata")
    eng = Engine(0)
    eng.load_edm(eargs, synth.synth_edm_state_dict(eargs, F, seed=1))
    eng.load_predictor(pargs, synth.synth_predictor_state_dict(pargs, F, 5, seed=2))
    for guided in (True, False):
        kw = dict(target_w=w, scale=0.6) if guided else {}
        eng.sample(nm, em, seed=1, **kw)
        t0 = time.perf_counter()
        x, h, d = eng.sample(nm, em, seed=2, **kw)
        dt = time.perf_counter() - t0
        print(f"sin_embedding={se} guided={guided}: B={B} T={T} {dt*1e3:.1f} ms  -> {B * T / dt / 1000:.1f} k molecule-steps/s "
              f"(= {B / (dt * 1000 / T):.1f} molecules/s at T=1000)  waves={eng.kernel_variant()[1]} nan={d['nan_count']}", flush=True)
    eng.close()
