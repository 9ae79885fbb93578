"""Scaling curve of the C++/OpenMP CPU baseline (oracle/gaudi_cpu.cpp) on this host: guided C3 steps (B = 256, N = 11, default
architectures) for several thread counts x molecules per thread-owned group.  Run on the GPU box through gpurun; the output
is kept under profiles/ (the stated baseline should be the host's best)."""
import os
import subprocess
import sys

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
for grp in (1, 2, 4, 8):
    port.set_group(grp)
    one(T-1)
    t0=time.time(); n=0
    while n<3: one(T-2-n); n+=1
    per=(time.time()-t0)/n
    print("threads", port.threads, "molecules per group", port.group_for(B), ":", round(per*1e3,1), "ms/step ->", round(B/(per*T),3), "guided mol/s (1000-step)", flush=True)
'''
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket|NUMA node\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
def cgroup(name):
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
        try:
            return open(os.path.join(base, name)).read().strip().replace("\n", " | ")
        except OSError:
            pass
    return "n/a"


# is the process CPU-quota limited?  (a container may SEE 256 hardware threads and be allowed the time of far fewer)
print("cgroup cpu.max:", cgroup("cpu.max"), "| cpu.cfs_quota_us:", cgroup("cpu.cfs_quota_us"), "| cpu.cfs_period_us:", cgroup("cpu.cfs_period_us"))
print("cgroup cpu.stat before:", cgroup("cpu.stat"))
print("sched_getaffinity:", len(os.sched_getaffinity(0)), "cpus; nproc:", subprocess.run("nproc", shell=True, capture_output=True, text=True).stdout.strip())
hw = os.cpu_count() or 8
for th in sorted({hw, max(1, hw // 2), max(1, hw // 4), max(1, hw // 8), 16}, reverse=True):
    if th > hw:
        continue
    env = dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND="spread", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr.strip()[-300:])
print("cgroup cpu.stat after:", cgroup("cpu.stat"))
