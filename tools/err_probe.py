import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaudi_amd import synth
from gaudi_amd.engine import Engine
from oracle import gaudi_oracle as O
from tests.helpers import max_norm_err, elem_err
from tests.test_gpu_fullsize import _z
ds, nodes, w = "cata", [11, 11, 7, 11, 4, 9], np.array([0, -1, 0, 0, 0], np.float32)
F = 1
eargs = synth.edm_args(dataset=ds, diffusion_steps=1000); pargs = synth.pred_args(dataset=ds)
esd = synth.synth_edm_state_dict(eargs, F, seed=0, amplify_coord=True); psd = synth.synth_predictor_state_dict(pargs, F, 5, seed=1, amplify_coord=True)
nm, em = O.build_masks(nodes, max(nodes), False)
z = _z(nm, F, 3); eps = np.random.default_rng(4).standard_normal(z.shape).astype(np.float32)
gamma = O.gamma_table("polynomial_2", 1000, 1e-5)
want = {s: O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6) for s in (999, 400, 0)}
want64 = {s: O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6, dtype=np.float64) for s in (999,)}
for wv in ("8", "4"):
    os.environ["GAUDI_WAVES"] = wv
    eng = Engine(0); eng.load_edm(eargs, esd); eng.load_predictor(pargs, psd)
    for s in (999, 400, 0):
        got = eng.step(s, z, nm, em, eps, target_w=w, scale=0.6)
        d = np.abs(got - want[s]); k = np.unravel_index(np.argmax(d / (np.abs(want[s]) + 1e-5 * np.abs(want[s]).max())), d.shape)
        print("waves", wv, "s", s, "maxnorm %.2e elem %.2e" % (max_norm_err(got, want[s]), elem_err(got, want[s])), "worst elem", k, got[k], want[s][k], "max|b|", np.abs(want[s]).max())
        if s in want64: print("   vs fp64 oracle: maxnorm %.2e elem %.2e ; fp32 oracle vs fp64: %.2e %.2e" % (max_norm_err(got, want64[s]), elem_err(got, want64[s]), max_norm_err(want[s], want64[s]), elem_err(want[s], want64[s])))
    eng.close()
wantu = {s: O.step_unguided(esd, eargs, gamma, s, z, nm, em, eps) for s in (999, 400, 0)}
wantu64 = O.step_unguided(esd, eargs, gamma, 999, z, nm, em, eps, dtype=np.float64)
for wv in ("8", "4"):
    os.environ["GAUDI_WAVES"] = wv
    eng = Engine(0); eng.load_edm(eargs, esd)
    for s in (999, 400, 0):
        got = eng.step(s, z, nm, em, eps)
        print("unguided waves", wv, "s", s, "maxnorm %.2e elem %.2e" % (max_norm_err(got, wantu[s]), elem_err(got, wantu[s])))
        if s == 999: print("   vs fp64: %.2e %.2e; fp32 oracle vs fp64 %.2e %.2e" % (max_norm_err(got, wantu64), elem_err(got, wantu64), max_norm_err(wantu[s], wantu64), elem_err(wantu[s], wantu64)))
    eng.close()
