"""Debug helper: batch (packed) vs one-molecule launches on the random-graph case of test_guided_steps_are_reproducible."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaudi_amd import synth
from gaudi_amd.engine import Engine
he, hp, S = 48, 40, 2
rng = np.random.default_rng(he * 7 + hp)
F, N, B, T = 3, 9, 5, 8
eargs = synth.edm_args(nf=he, n_layers=2, inv_sublayers=S, diffusion_steps=T)
pargs = synth.pred_args(nf=hp, n_layers=3)
esd = synth.synth_edm_state_dict(eargs, F, seed=71, amplify_coord=True)
psd = synth.synth_predictor_state_dict(pargs, F, 4, seed=72, amplify_coord=True)
n_live = rng.integers(2, N + 1, size=B)
nm = (np.arange(N)[None, :] < n_live[:, None]).astype(np.float32)[:, :, None]
em = ((rng.random((B, N, N)) < 0.6) * (1 - np.eye(N))[None]).astype(np.float32) * nm * nm.transpose(0, 2, 1)
z = rng.standard_normal((B, N, 3 + F)).astype(np.float32) * nm
z[:, :, :3] -= z[:, :, :3].sum(1, keepdims=True) / nm.sum(1, keepdims=True) * nm
eps = rng.standard_normal(z.shape).astype(np.float32)
w = np.array([0.5, -1.0, 0.25, 0.0], np.float32)
print("n_live", n_live, "edges", em.reshape(B, -1).sum(1))
eng = Engine(0); eng.load_edm(eargs, esd); eng.load_predictor(pargs, psd)
t = np.full(B, 1.0, np.float32)
for name, f in (("unguided", lambda zz, n, e, ep: eng.step(7, zz, n, e, ep)), ("guided", lambda zz, n, e, ep: eng.step(7, zz, n, e, ep, target_w=w, scale=0.8)),
                ("guided_s0", lambda zz, n, e, ep: eng.step(0, zz, n, e, ep, target_w=w, scale=0.8))):
    ref = f(z, nm, em, eps)
    d = []
    for b in range(B):
        one = f(z[b:b + 1], nm[b:b + 1], em[b:b + 1], eps[b:b + 1])
        d.append(float(np.abs(one[0] - ref[b]).max() / max(np.abs(ref[b]).max(), 1e-30)))
    print(name, " ".join(f"{v:.1e}" for v in d), flush=True)
pg = eng.predictor_grad(z, t, nm, em, np.broadcast_to(w, (B, 4)).copy())
ref = eng.step(7, z, nm, em, eps)
one = eng.step(7, z[0:1], nm[0:1], em[0:1], eps[0:1])
print("diff mask (rows = nodes, cols = x,y,z,h0,h1,h2):")
print((one[0] != ref[0]).astype(int))
print(one[0][:4] - ref[0][:4])
os.environ["GAUDI_PACK"] = "0"
e2 = Engine(0); e2.load_edm(eargs, esd); e2.load_predictor(pargs, psd)
r2 = e2.step(7, z, nm, em, eps)
print("unpacked batch == single:", np.array_equal(r2[0], one[0]), " unpacked batch == packed batch:", np.array_equal(r2, ref))
# swap molecule order so that molecule 0 leads its group
perm = [1, 0, 2, 3, 4]
r3 = eng.step(7, z[perm], nm[perm], em[perm], eps[perm])
print("after swapping 0<->1: mol0 equal single:", np.array_equal(r3[1], one[0]), "mol1 equal:", np.array_equal(r3[0], r2[1]))
# --- union graph by hand through the UNPACKED phi entry point: h rows of molecule 0 at slots 5..8 vs alone at 0..3
nmu = np.zeros((1, N), np.float32); emu = np.zeros((1, N, N), np.float32); zu = np.zeros((1, N, 3 + F), np.float32)
nmu[0, :5] = 1; nmu[0, 5:9] = 1
emu[0, :5, :5] = em[1, :5, :5]; emu[0, 5:9, 5:9] = em[0, :4, :4]
zu[0, :5] = z[1, :5]; zu[0, 5:9] = z[0, :4]
pu = e2.phi(zu, 0.5, nmu, emu)
p0 = e2.phi(z[0:1], 0.5, nm[0:1], em[0:1])
p1 = e2.phi(z[1:2], 0.5, nm[1:2], em[1:2])
print("union phi h-part, mol0 rows equal:", np.array_equal(pu[0, 5:9, 3:], p0[0, :4, 3:]), " mol1 rows equal:", np.array_equal(pu[0, :5, 3:], p1[0, :5, 3:]))
print((pu[0, 5:9, 3:] != p0[0, :4, 3:]).astype(int))
# same with the fp32 edge math and the 4-wave kernels
for env in ({"GAUDI_EDGE_MATH": "fp32"}, {"GAUDI_WAVES": "4"}):
    os.environ.update(env)
    e3 = Engine(0); e3.load_edm(eargs, esd)
    a, b_ = e3.phi(zu, 0.5, nmu, emu), e3.phi(z[0:1], 0.5, nm[0:1], em[0:1])
    print(env, "mol0 h rows equal:", np.array_equal(a[0, 5:9, 3:], b_[0, :4, 3:]))
    for k in env: os.environ.pop(k)
    e3.close()
