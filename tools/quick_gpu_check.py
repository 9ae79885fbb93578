"""Ad-hoc GPU check: parity error magnitudes + first timings (not part of the test suite)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaudi_amd import synth
from gaudi_amd.engine import Engine
from oracle import gaudi_oracle as O
from tests.helpers import cfg_of, edm_from_cfg, rel_err

g = dict(np.load("tests/golden/g3_phi.npz"))
for name in ["cata_tiny_amp", "cata_full", "hetro_full_amp"]:
    cfg = cfg_of(g, name)
    args, sd = edm_from_cfg(cfg)
    eng = Engine(0); eng.load_edm(args, sd)
    eps = eng.phi(g[name+"_z"], g[name+"_t"][:,0], g[name+"_node_mask"], g[name+"_edge_mask"])
    print(name, "phi rel err vs reference:", rel_err(eps, g[name+"_eps"]))
    eng.close()

# C2-shaped timing
B = int(os.environ.get("B", 256)); T = 1000
eargs = synth.edm_args()
esd = synth.synth_edm_state_dict(eargs, 1, seed=0)
eng = Engine(0); eng.load_edm(eargs, esd)
nm, em = O.build_masks([11]*B, 11, False)
z = np.random.default_rng(0).standard_normal((B, 11, 4)).astype(np.float32)
for rep in range(3):
    t0 = time.time(); eng.phi(z, 0.5, nm, em); t1 = time.time()
    print("phi call wall ms", (t1-t0)*1e3)
eng.profile_reset(True)
eng.set_steps_per_launch(int(os.environ.get("SPL", 25)))
eps = np.zeros((B,11,4), np.float32)
# time 50 steps by running steps 49..0 via sample with T... use gaudi_step repeatedly is slow; use sample on a T=50 model
eargs50 = synth.edm_args(diffusion_steps=50)
eng2 = Engine(0); eng2.load_edm(eargs50, synth.synth_edm_state_dict(eargs50, 1, seed=0))
eng2.profile_reset(True)
t0 = time.time(); x, h, d = eng2.sample(nm, em, seed=1); t1 = time.time()
n, ms, steps = eng2.profile_get()
print(f"T=50 sample B={B}: wall {t1-t0:.3f}s kernel {ms:.1f} ms over {n} launches, {steps} steps -> {ms/max(steps,1):.3f} ms/step(+decode)", d)
