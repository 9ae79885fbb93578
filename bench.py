#!/usr/bin/env python3
"""Headline benchmark: guided molecules/sec (1000-step) on MI355X.

A "step" (--steps K) is ONE complete pass of the hot path over one batch: a full
``sample_guidance`` call = 1000 guided reverse-diffusion steps + final decode for B molecules
(BASELINE.json config C3: cc-PBH 11-ring, batch 256, HOMO-LUMO-gap guidance, default architectures,
synthetic seeded weights, on-device Philox noise).  With --gpus N the driver launches N ranks
(torch.distributed.run); each rank samples its own 256 molecules (weak scaling, noise keyed by the
global sample index) and ONE all_gather over RCCL collects the results at the end of every step.

Prints one JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the fused
per-molecule sampler kernel, bound = fp32 matrix cores) and `cpu_baseline` (the numpy oracle timed on
this host, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* = vector fp32 peak
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sample_guidance calls (1000 reverse steps each)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="molecules per GPU (default 256; 1024 for c4)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "stability"],
                    help="c2 = unguided cata, c3 = gap-guided cata (headline), c4 = hetero mixed 3-10 rings, multi-objective; "
                         "stability = the graph-of-rings stability kernel that follows sampling (SURVEY 8f rank 1)")
    ap.add_argument("--molecules", type=int, default=262144, help="stability workload: molecules per call")
    ap.add_argument("--dataset", default="cata", choices=["cata", "hetro"], help="stability workload: geometry tables")
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--steps-per-launch", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(eargs, pargs, esd, psd, guided, T, B):
    """The numpy oracle (kind "port") on this host's cores over a bounded sample of the SAME workload shape:
    a few reverse steps of the full batch (B molecules, N=11), extrapolated to T steps + decode."""
    from oracle import gaudi_oracle as O  # baseline leg only
    nm, em = O.build_masks([11] * B, 11, False)
    rng = np.random.default_rng(0)
    z = O._combined_noise(rng.standard_normal((B, 11, 4)).astype(np.float32), nm)
    gamma = O.gamma_table(eargs["diffusion_noise_schedule"], T, eargs["diffusion_noise_precision"])
    w = O.target_max_gap_weights(5)

    def one(s):
        eps = rng.standard_normal((B, 11, 4)).astype(np.float32)
        if guided:
            return O.step_guided(esd, eargs, psd, pargs, gamma, s, z, nm, em, eps, w, 0.6)
        return O.step_unguided(esd, eargs, gamma, s, z, nm, em, eps)

    one(T - 1)  # warm-up
    n_steps, t0 = 0, time.time()
    while n_steps < 2 or (time.time() - t0 < 12.0 and n_steps < 20):
        one(T - 2 - n_steps)
        n_steps += 1
    per_step = (time.time() - t0) / n_steps
    total = per_step * T * (1.0 + (1.0 / T) * (0.3 if guided else 1.0))  # + decode = one EDM evaluation
    return dict(value=B / total, unit="molecules/s", cores=os.cpu_count(), kind="port",
                sample=f"numpy oracle (BLAS threads = all {os.cpu_count()} cores), B={B} x {n_steps} "
                       f"{'guided' if guided else 'unguided'} reverse steps at N=11, extrapolated x{T} + decode; "
                       f"{per_step * 1e3:.0f} ms/step")


def bench_stability(a):
    """Side workload: gaudi_check_stability on replicated golden molecules (one step = one call over `--molecules`).
    Same JSON contract; the kernel is HBM-side by its algorithmic bytes but bound by its serial sections."""
    from gaudi_amd import analyze
    from gaudi_amd.engine import Engine
    g = np.load(os.path.join(ROOT, "tests", "golden", "g11_stability.npz"))
    ds, M = a.dataset, a.molecules
    idx = np.arange(M) % len(g[f"{ds}_n"])
    X = np.ascontiguousarray(g[f"{ds}_x"][idx])
    T = np.ascontiguousarray(np.maximum(g[f"{ds}_types"][idx], 0).astype(np.int32))
    nn = np.ascontiguousarray(g[f"{ds}_n"][idx].astype(np.int32))
    eng = Engine(0)
    want = g[f"{ds}_flags"].astype(bool)[idx]
    for _ in range(max(a.warmup, 1)):
        flags = analyze.check_stability_batch(X, T, nn, 0.1, ds, engine=eng)
    assert np.array_equal(flags, want), "stability flags differ from the reference's (g11 fixture)"
    eng.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        analyze.check_stability_batch(X, T, nn, 0.1, ds, engine=eng)
    wall = (time.perf_counter() - t0) / a.steps
    n_launch, kernel_ms = eng.stability_profile_get()
    kernel_s = kernel_ms / 1e3 / max(n_launch, 1)
    bytes_per_mol = X.shape[1] * 3 * 4 + X.shape[1] * 4 + 4 + 5
    out = dict(metric="stability-checked molecules/sec", value=M / wall, unit="molecules/s", n_gpus=1, steps=a.steps,
               warmup=max(a.warmup, 1), ms_per_step=wall * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32", data="synthetic (g11 fixture molecules replicated)",
               config=dict(workload=f"graph-of-rings stability check, {ds}, {M} molecules per call, N={X.shape[1]}"),
               roofline=dict(bound="hbm", achieved=M * bytes_per_mol / kernel_s / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s",
                             frac=M * bytes_per_mol / kernel_s / 1e9 / PEAK_HBM_GBPS, traffic=None,
                             kernel="stability_kernel", avg_launch_ms=kernel_s * 1e3, launches=n_launch,
                             algorithmic_bytes_per_molecule=bytes_per_mol, kernel_molecules_per_s=M / kernel_s))
    if not a.no_cpu_baseline:
        from oracle import stability_oracle as S  # baseline leg only
        m, t0 = 0, time.perf_counter()
        while m < M and (m < 64 or time.perf_counter() - t0 < 10.0):
            S.check_stability(X[m, : nn[m]], T[m, : nn[m]], dataset=ds)
            m += 1
        out["cpu_baseline"] = dict(value=m / (time.perf_counter() - t0), unit="molecules/s", cores=1, kind="port",
                                   sample=f"numpy/Python stability oracle on the first {m} molecules of the same batch")
    print(json.dumps(out))


def main():
    a = parse()
    if a.workload == "stability":
        return bench_stability(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        # convenience: relaunch under torch.distributed.run as a CHILD process (never exec after GPU init)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    from gaudi_amd import dist as gdist
    from gaudi_amd import flops, synth
    from gaudi_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sampler has no CPU fallback")
    # GAUDI_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing test on a 1-GPU box); the real
    # multi-GPU run uses RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("GAUDI_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    guided = a.workload in ("c3", "c4")
    hetero = a.workload == "c4"
    T, K = a.diffusion_steps, 5
    B = a.batch or (1024 if hetero else 256)
    N, F = (20, 12) if hetero else (11, 1)
    eargs = synth.edm_args(diffusion_steps=T, dataset="hetro" if hetero else "cata")
    pargs = synth.pred_args(dataset="hetro" if hetero else "cata")
    esd = synth.synth_edm_state_dict(eargs, F, seed=0)
    psd = synth.synth_predictor_state_dict(pargs, F, K, seed=1)
    eng = Engine(gpu)
    eng.load_edm(eargs, esd)
    if guided:
        eng.load_predictor(pargs, psd)
    eng.set_steps_per_launch(a.steps_per_launch)
    if hetero:
        # PASs-like batch: 3..10 rings drawn uniformly (seed 1), orientation nodes -> 6..20 graph nodes, N = 20
        from gaudi_amd.sampling_edm import build_masks
        rings = np.random.default_rng(1 + rank).integers(3, 11, size=B)
        nm3, em_flat, _ = build_masks(rings, 10, True)
        nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
        live_edges, live_nodes = float(em.sum() / B), float(nm.sum() / B)
    else:
        nm = np.ones((B, N), np.float32)  # 11-ring cata molecules: every node live (sampling_edm.py:176-186)
        em = np.broadcast_to(1.0 - np.eye(N, dtype=np.float32), (B, N, N)).copy()
        live_edges, live_nodes = float(N * (N - 1)), float(N)
    tw = None
    if guided:
        tw = np.zeros(K, np.float32)
        if hetero:
            tw[0], tw[2], tw[3] = 3.0, 1.0, 1.0  # target_function_opv with mean=0, std=1 (generation_guidance.py:205-211)
        else:
            tw[1] = -1.0  # target_function_max_gap: -pred[:,1]  (generation_guidance.py:200-203)

    def one_pass(it):
        x, h, diag = eng.sample(nm, em, seed=1234 + it, sample_offset=rank * B, std=1.0, target_w=tw, scale=0.6)
        if world > 1:
            x, h = gdist.gather_to_all(x, h, B * world, N, F, device=dev if backend == "nccl" else None)
        return x, h, diag

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(a.warmup):
        one_pass(-1 - it)
    eng.profile_reset(True)
    sync()
    t0 = time.perf_counter()
    for it in range(a.steps):
        x, h, diag = one_pass(it)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_launch, kern_ms, steps_done = eng.profile_get()
    assert x.shape[0] == B * world and np.isfinite(x).all()

    if rank == 0:
        mols = a.steps * B * world
        value = mols / dt
        # ---- roofline of the dominant kernel (sampler_kernel<192,208>: EDM + predictor fwd/bwd + update)
        f_written = flops.step_flops_as_written(N, F, eargs, pargs if guided else None, K)
        f_useful = flops.step_flops_useful(live_edges, live_nodes, F, eargs, pargs if guided else None, K)
        evals = steps_done + a.steps * (0.3 if guided else 1.0)  # decode pass = one extra EDM evaluation
        avg_launch_ms = kern_ms / max(n_launch, 1)
        per_launch_flops = f_written * B * evals / max(n_launch, 1)
        achieved = per_launch_flops / (avg_launch_ms * 1e-3) / 1e12
        wbytes = 4 * (sum(v.size for v in esd.values()) + (2 * sum(v.size for v in psd.values()) if guided else 0))
        # predictor stash per molecule-step: node part (P, Q, npre, x) + edge part (v, cpre of every 16-edge tile)
        stash = 4 * pargs["n_layers"] * ((3 * N * 208 + 4 * N) + 4 * 32 * 208 * 2) if guided else 0
        hbm_bytes_launch = flops.step_bytes_fused(B, N, F, 0, stash) * steps_done / max(n_launch, 1) + wbytes
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and B == 256 and a.steps_per_launch == 25:  # counters were collected on this shape
            try:
                traffic = json.load(open(pmc)).get(f"{a.workload}_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": ("guided" if guided else "unguided") + f" molecules/sec ({T}-step)",
            "value": value, "unit": "molecules/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded default-init weights, on-device Philox noise)",
            "config": {"workload": {"c3": f"C3: cc-PBH 11-ring, batch={B}/GPU, {T} steps, HOMO-LUMO-gap guidance (scale 0.6)",
                                    "c2": f"C2: cc-PBH 11-ring, batch={B}/GPU, {T} steps, unconditional EDM",
                                    "c4": f"C4: PASs-like hetero, 3-10 rings (6-20 graph nodes, N=20), batch={B}/GPU, {T} steps, "
                                          "multi-objective (OPV) guidance"}[a.workload],
                       "global_batch": B * world, "n_nodes": N, "diffusion_steps": T,
                       "edm": "nf=192,n_layers=9", "predictor": "nf=196,n_layers=12" if guided else None,
                       "parallelism": f"sample-sharded x{world}, one RCCL all_gather per call",
                       "steps_per_launch": a.steps_per_launch},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                         "kernel": "sampler_kernel<192,208>" if guided else "sampler_kernel<192,0>",
                         "live_edges_per_molecule": live_edges,
                         "avg_launch_ms": avg_launch_ms, "launches": n_launch,
                         "flops_basis": "as-written reference FLOPs (SURVEY 8d): %.3f GFLOP per molecule-step" % (f_written / 1e9),
                         "useful_tflops_factorised": f_useful * B * evals / (kern_ms * 1e-3) / 1e12,
                         "hbm_algorithmic_gbps": hbm_bytes_launch / (avg_launch_ms * 1e-3) / 1e9,
                         "hbm_frac": hbm_bytes_launch / (avg_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS},
            "diag": diag,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth.edm_args(diffusion_steps=T), synth.pred_args(),
                                                synth.synth_edm_state_dict(synth.edm_args(diffusion_steps=T), 1, seed=0),
                                                synth.synth_predictor_state_dict(synth.pred_args(), 1, K, seed=1), guided, T,
                                                256)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
