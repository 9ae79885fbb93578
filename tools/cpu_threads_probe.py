import os, sys, time, subprocess, json
code = r'''
import os, time, numpy as np, sys
sys.path.insert(0, os.getcwd())
from oracle import build_cpu
from oracle import gaudi_oracle as O
from gaudi_amd import synth
T=1000
eargs, pargs = synth.edm_args(diffusion_steps=T), synth.pred_args()
esd, psd = synth.synth_edm_state_dict(eargs, 1, seed=0), synth.synth_predictor_state_dict(pargs, 1, 5, seed=1)
port = build_cpu.CpuPort(); port.load_edm(eargs, esd); port.load_predictor(pargs, psd)
B=256
nm, em = O.build_masks([11]*B, 11, False)
rng=np.random.default_rng(0)
z=O._combined_noise(rng.standard_normal((B,11,4)).astype(np.float32), nm)
gamma=O.gamma_table("polynomial_2", T, 1e-5); w=O.target_max_gap_weights(5)
def one(s):
    eps=rng.standard_normal((B,11,4)).astype(np.float32)
    return port.step(O.step_coefficients(gamma,s,s+1), np.float32(np.float32(s+1)/np.float32(T)), z, nm, em, eps, target_w=w, scale=0.6)
one(T-1)
t0=time.time(); n=0
while n<4: one(T-2-n); n+=1
per=(time.time()-t0)/n
print(port.threads, round(per*1e3,1), "ms/step ->", round(B/(per*T),4), "mol/s")
'''
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket'", shell=True, capture_output=True, text=True).stdout)
for th in (128, 64, 32):
    env = dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND="spread", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(th, r.stdout.strip(), r.stderr.strip()[-200:])
