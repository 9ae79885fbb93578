#!/usr/bin/env python3
"""Headline benchmark: guided molecules/sec (1000-step) on MI355X.

A "step" (--steps K) is ONE complete pass of the hot path over one batch: a full
``sample_guidance`` call = 1000 guided reverse-diffusion steps + final decode for B molecules
(BASELINE.json config C3: cc-PBH 11-ring, batch 256, HOMO-LUMO-gap guidance, default architectures,
synthetic seeded weights, on-device Philox noise).  With --gpus N > 1 the driver launches N ranks
(torch.distributed.run) and the workload is BASELINE config C5: each rank samples its own 1024 molecules
(8192 over 8 GPUs; weak scaling, noise keyed by the global sample index) and ONE all_gather over RCCL
collects the results at the end of every step.

Prints one JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the fused
per-molecule sampler kernel, bound = fp32 matrix cores, fraction = ISSUED matrix FLOPs / time / peak),
`secondary` (short C2, C4 and C3-at-1024 passes, N=1 only) and `cpu_baseline` (the C++/OpenMP restatement
oracle/gaudi_cpu.cpp timed on this host's cores, N=1 only).
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
NOMINAL_CLOCK_MHZ = 2400.0  # the clock the guide's peak figures are quoted at


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sample_guidance calls (1000 reverse steps each)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None,
                    help="molecules per GPU (default: 256 = C3/C2 at one GPU; 1024 for c4 and for --gpus > 1 = C5's per-GPU shard)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c4x", "stability"],
                    help="c2 = unguided cata, c3 = gap-guided cata (headline), c4 = hetero mixed 3-10 rings, multi-objective; c4x = the same at 6-20 rings (12-40 graph nodes: BASELINE config 4 read literally; V4G kernels, not in the default run); "
                         "stability = the graph-of-rings stability kernel that follows sampling (SURVEY 8f rank 1)")
    ap.add_argument("--molecules", type=int, default=262144, help="stability workload: molecules per call")
    ap.add_argument("--dataset", default="cata", choices=["cata", "hetro"], help="stability workload: geometry tables")
    ap.add_argument("--nodes", type=int, default=11, help="c2 / c3: rings of the cata molecules (BASELINE: 11; fully connected graphs)")
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--steps-per-launch", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the untimed full-batch step against the CPU restatement")
    ap.add_argument("--closure", default=None, choices=["linear", "nonlinear"],
                    help="c3 only, diagnostic: run the timed calls through sampling_edm.sample_guidance with a reference-form closure")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short C2 / C4 / C3-at-1024 passes after the headline")
    ap.add_argument("--dist", action="store_true",
                    help="N = 1 only: route the run through the SAME distributed code as N > 1 (init_process_group('nccl', "
                         "world_size=1), the RCCL all_gather of every call, barrier + MAX all_reduce around the timed region)")
    return ap.parse_args()


def cpu_quota():
    """CPUs' worth of time this process may use (cgroup v2 cpu.max / v1 cfs quota), or None: a container can SEE 256 hardware
    threads and be allowed the time of 16 -- more threads than that only add throttling."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(round(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, int(round(q / p)))
    except (OSError, ValueError):
        return None


def cpu_baseline(eargs, pargs, esd, psd, guided, T, B):
    """The C++/OpenMP restatement (oracle/gaudi_cpu.cpp, kind "port": the reference's arithmetic as written -- concat ->
    Linear over the dense N x N edge set -- one molecule per thread, AVX-512 / AVX2 GEMM micro-kernels) on this host's cores,
    over a bounded sample of the SAME workload shape: a few reverse steps of the full batch (B molecules, N = 11),
    extrapolated to T steps + decode.  BASELINE.md section 3: >= 5 timed steps, all cores."""
    from oracle import build_cpu  # baseline leg only
    from oracle import gaudi_oracle as O
    port = build_cpu.CpuPort()
    port.load_edm(eargs, esd)
    if guided:
        port.load_predictor(pargs, psd)
    nm, em = O.build_masks([11] * B, 11, False)
    rng = np.random.default_rng(0)
    z = O._combined_noise(rng.standard_normal((B, 11, 4)).astype(np.float32), nm)
    gamma = O.gamma_table(eargs["diffusion_noise_schedule"], T, eargs["diffusion_noise_precision"])
    w = O.target_max_gap_weights(5)

    def one(s):
        eps = rng.standard_normal((B, 11, 4)).astype(np.float32)
        return port.step(O.step_coefficients(gamma, s, s + 1), np.float32(np.float32(s + 1) / np.float32(T)), z, nm, em, eps,
                         target_w=w if guided else None, scale=0.6)

    one(T - 1)  # warm-up
    # A thread owns a GROUP of molecules and runs them layer by layer (one GEMM per Linear over the group's rows): a weight
    # matrix is read once per group, so the rate follows the cores rather than the memory system (one molecule at a time
    # re-streams the 25-54 MB weight set per molecule: 0.99 mol/s on 16 threads, 0.64 on 128, round 3).  Probe thread counts x
    # group sizes (one step each) and time the sample with the best; `cores` reports the threads actually used.
    hw = port.threads
    quota = cpu_quota()
    cand = {hw, max(1, hw // 2), max(1, hw // 4), max(1, hw // 8)}
    if quota is not None:  # the host gives this process `quota` CPUs' worth of time: probe around that, not around the thread count
        cand = {min(hw, quota), min(hw, 2 * quota), max(1, quota // 2)}
    best_n, best_g, best_t, probes = hw, 1, None, {}
    for n_thr in sorted(cand, reverse=True):
        for grp in (4, 2, 1):
            port.set_threads(n_thr)
            port.set_group(grp)
            dt1 = None
            for _ in range(2):  # the faster of two steps (a single step is noisy on a shared host)
                t1 = time.time()
                one(T - 1)
                dt1 = time.time() - t1 if dt1 is None else min(dt1, time.time() - t1)
            probes[f"{n_thr}x{port.group_for(B)}"] = round(B / dt1 / T, 3)
            if best_t is None or dt1 < best_t:
                best_n, best_g, best_t = n_thr, grp, dt1
    port.set_threads(best_n)
    port.set_group(best_g)
    group_used = port.group_for(B)
    n_steps, t0 = 0, time.time()
    while n_steps < 5 or (time.time() - t0 < 20.0 and n_steps < 100):
        one(T - 2 - n_steps)
        n_steps += 1
    per_step = (time.time() - t0) / n_steps
    total = per_step * T * (1.0 + (1.0 / T) * (0.3 if guided else 1.0))  # + decode = one EDM evaluation
    ref = 0.058 if guided else 0.175
    threads, isa = best_n, port.isa
    port.close()
    return dict(value=B / total, unit="molecules/s", cores=threads, kind="port",
                sample=f"C++/OpenMP restatement oracle/gaudi_cpu.cpp ({isa} GEMM micro-kernel, {threads} threads x groups of {group_used} "
                       f"molecules per thread = the fastest of {sorted(cand, reverse=True)} threads x groups of <= 4 / 2 / 1 on this "
                       f"host: {hw} hardware threads visible" + (f", cgroup CPU quota {quota} CPUs" if quota is not None else "")
                       + f"), B={B} x {n_steps} {'guided' if guided else 'unguided'} reverse steps at N=11 after 1 warm-up, "
                       f"extrapolated x{T} + decode; {per_step * 1e3:.0f} ms/step.  Cross-check (BASELINE.md section 2): the "
                       f"reference's own PyTorch-CPU path measured {ref} molecules/s on the 8-core build container for this "
                       f"workload, where this port measures 0.141 guided / 0.333 unguided; speedup_vs_cpu_baseline divides by the "
                       f"larger of the port's figure on this host and the reference's 8-core figure",
                per_core=B / total / max(threads, 1), reference_torch_cpu_8core=ref, reference_torch_cpu_per_core=ref / 8,
                hardware_threads=hw, cpu_quota=quota, molecules_per_group=group_used,
                probe_mol_per_s={"note": "threads x molecules per group -> molecules/s of one probe step", **probes})


def parity_gate(eng, eargs, pargs, esd, psd, nm, em, tw, T, tol=1e-4):
    """One teacher-forced reverse step (s = T/2) of the FULL batch on the engine the timed loop is about to use, checked
    molecule by molecule against the CPU restatement (oracle/: the C++ port, itself pinned to the reference's goldens; the
    numpy oracle on the first 8 molecules when the port cannot be built here).  Untimed; a failing gate exits non-zero --
    a fast kernel with wrong results must not print a bench line."""
    from oracle import gaudi_oracle as O  # checker only
    B, N = nm.shape
    D = 3 + (np.asarray(esd["dynamics.egnn.embedding.weight"]).shape[1] - 1)
    rng = np.random.default_rng(77)
    z = O._combined_noise(rng.standard_normal((B, N, D)).astype(np.float32), nm[:, :, None])
    eps = rng.standard_normal((B, N, D)).astype(np.float32)
    s = T // 2
    gamma = O.gamma_table(eargs["diffusion_noise_schedule"], T, eargs["diffusion_noise_precision"])
    got = eng.step(s, z, nm, em, eps, target_w=tw, scale=0.6)
    checker, rows = "oracle/gaudi_cpu.cpp (C++ port)", slice(0, B)
    try:
        from oracle import build_cpu
        if not build_cpu.cpu_ok():
            raise RuntimeError("host CPU lacks AVX2/FMA")
        port = build_cpu.CpuPort()
        port.load_edm(eargs, esd)
        if tw is not None:
            port.load_predictor(pargs, psd)
        want = port.step(O.step_coefficients(gamma, s, s + 1), np.float32(np.float32(s + 1) / np.float32(T)), z, nm, em, eps,
                         target_w=tw, scale=0.6)
        port.close()
    except Exception as exc:  # no g++ / no AVX2 on this host: the numpy oracle on a few molecules
        rows = slice(0, min(B, 8))
        checker = f"oracle/gaudi_oracle.py (numpy, first {rows.stop} molecules; C++ port unavailable: {exc})"
        if tw is None:
            want = O.step_unguided(esd, eargs, gamma, s, z[rows], nm[rows][:, :, None], em[rows], eps[rows])
        else:
            want = O.step_guided(esd, eargs, psd, pargs, gamma, s, z[rows], nm[rows][:, :, None], em[rows], eps[rows], tw, 0.6)
    g, w = got[rows].reshape(want.shape[0], -1), np.asarray(want).reshape(want.shape[0], -1)
    per_mol = np.abs(g - w).max(1) / np.maximum(np.abs(w).max(1), 1e-30)
    rel = float(per_mol.max())
    leak = bool(np.any(got[nm == 0] != 0))
    gate = {"rel_err": rel, "tol": tol, "passed": bool(rel < tol and not leak and np.isfinite(got).all()),
            "what": f"one teacher-forced {'guided' if tw is not None else 'unguided'} reverse step (s = {s}) of the full batch "
                    f"({B} molecules) on the timed engine, max over molecules of max|HIP - CPU| / max|CPU|",
            "checker": checker, "kernel_waves": eng.kernel_variant()[1], "edge_math": eng.edge_math()[1]}
    return gate


def bench_stability(a, emit):
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
    emit(out)


def graph_meta(nm, em):
    """The library's own live-edge metadata (32-edge passes per wave, node columns) for the issued-MFMA count."""
    import ctypes as C
    from gaudi_amd import _lib
    lib = _lib.load_library()
    B, N = nm.shape
    ew = C.c_int32()
    npairs = np.zeros((B, 4), np.int32)
    ncols = np.zeros(B, np.int32)
    i32 = C.POINTER(C.c_int32)
    rc = lib.gaudi_host_graph_meta(B, N, _lib.fptr(np.ascontiguousarray(nm, np.float32)),
                                   _lib.fptr(np.ascontiguousarray(em, np.float32)), C.byref(ew), None,
                                   npairs.ctypes.data_as(i32), None, None, None, 0, ncols.ctypes.data_as(i32))
    assert rc == 0, rc
    return npairs, ncols


def graph_meta8_packed(nm, em):
    """16-slot tiles and node columns per WORKGROUP of a sampling call of the 8-wave kernels: small molecules are packed
    into one workgroup as components of a disjoint graph (gaudi_host_pack_plan); G = B when nothing packs."""
    import ctypes as C
    from gaudi_amd import _lib
    lib = _lib.load_library()
    B, N = nm.shape
    G = C.c_int32()
    ntiles = np.zeros(B, np.int32)
    ncols = np.zeros(B, np.int32)
    i32 = C.POINTER(C.c_int32)
    rc = lib.gaudi_host_pack_plan(B, N, _lib.fptr(np.ascontiguousarray(nm, np.float32)), _lib.fptr(np.ascontiguousarray(em, np.float32)),
                                  C.byref(G), None, ntiles.ctypes.data_as(i32), ncols.ctypes.data_as(i32))
    assert rc == 0, rc
    return ntiles[:G.value], ncols[:G.value]


def graph_meta8(nm, em):
    """16-slot tiles per molecule and node columns of the 8-wave kernels."""
    import ctypes as C
    from gaudi_amd import _lib
    lib = _lib.load_library()
    B, N = nm.shape
    slots = C.c_int32()
    ntiles = np.zeros(B, np.int32)
    ncols = np.zeros(B, np.int32)
    i32 = C.POINTER(C.c_int32)
    rc = lib.gaudi_host_graph_meta8(B, N, _lib.fptr(np.ascontiguousarray(nm, np.float32)),
                                    _lib.fptr(np.ascontiguousarray(em, np.float32)), C.byref(slots), None,
                                    ntiles.ctypes.data_as(i32), None, None, None, None, None, 0, ncols.ctypes.data_as(i32))
    assert rc == 0, rc
    return ntiles, ncols


def run_workload(a, eng_cache, workload, B, steps, warmup, rank, world, dev, backend, T, K=5, edge_math=None, use_dist=False,
                 gate=False, closure=None):
    """Time `steps` complete sampling calls of one workload; returns the contract fields + roofline of its kernel.
    edge_math="fp32": a handle created with GAUDI_EDGE_MATH=fp32 (edge GEMMs on fp32 matrix instructions)."""
    import torch
    import torch.distributed as dist
    from gaudi_amd import dist as gdist
    from gaudi_amd import flops, synth
    from gaudi_amd.engine import Engine

    guided = workload in ("c3", "c4", "c4x")
    hetero = workload in ("c4", "c4x")
    max_rings = 20 if workload == "c4x" else 10
    N, F = (2 * max_rings, 12) if hetero else (a.nodes, 1)
    ds = "hetro" if hetero else "cata"
    eargs = synth.edm_args(diffusion_steps=T, dataset=ds)
    pargs = synth.pred_args(dataset=ds)
    key = (ds, guided, edge_math)
    if key not in eng_cache:
        saved = os.environ.get("GAUDI_EDGE_MATH")
        if edge_math is not None:
            os.environ["GAUDI_EDGE_MATH"] = edge_math  # read once, by gaudi_create
        try:
            eng = Engine(dev.index)
        finally:
            if edge_math is not None:
                if saved is None:
                    del os.environ["GAUDI_EDGE_MATH"]
                else:
                    os.environ["GAUDI_EDGE_MATH"] = saved
        esd = synth.synth_edm_state_dict(eargs, F, seed=0)
        eng.load_edm(eargs, esd)
        wfloats = sum(v.size for v in esd.values())
        if guided:
            psd = synth.synth_predictor_state_dict(pargs, F, K, seed=1)
            eng.load_predictor(pargs, psd)
            wfloats += 2 * sum(v.size for v in psd.values())  # + the transposed copies the reverse pass streams
        eng_cache[key] = (eng, 4 * wfloats, esd, psd if guided else None)
    eng, wbytes, esd_k, psd_k = eng_cache[key]
    eng.set_steps_per_launch(a.steps_per_launch)
    if hetero:
        # PASs-like batch: 3..10 rings drawn uniformly (seed 1), orientation nodes -> 6..20 graph nodes, N = 20
        from gaudi_amd.sampling_edm import build_masks
        rings = np.random.default_rng(1 + rank).integers(6 if workload == "c4x" else 3, max_rings + 1, size=B)
        nm3, em_flat, _ = build_masks(rings, max_rings, True)
        nm, em = nm3.reshape(B, N), em_flat.reshape(B, N, N)
    else:
        nm = np.ones((B, N), np.float32)  # 11-ring cata molecules: every node live (sampling_edm.py:176-186)
        em = np.broadcast_to(1.0 - np.eye(N, dtype=np.float32), (B, N, N)).copy()
    live_edges, live_nodes = float(em.sum() / B), float(nm.sum() / B)
    tw = None
    if guided:
        tw = np.zeros(K, np.float32)
        if hetero:
            tw[0], tw[2], tw[3] = 3.0, 1.0, 1.0  # target_function_opv with mean=0, std=1 (generation_guidance.py:205-211)
        else:
            tw[1] = -1.0  # target_function_max_gap: -pred[:,1]  (generation_guidance.py:200-203)

    gate_out = None
    if gate and rank == 0:
        gate_out = parity_gate(eng, eargs, pargs, esd_k, psd_k, nm, em, tw, T)
    closure_target = None
    if closure is not None:
        # The reference-form call: sampling_edm.sample_guidance(args, model, <closure over cond_predictor>, nodesxsample, scale)
        # (generation_guidance.py:198-211, sampling_edm.py:172-224) through the Python mirror, on this engine.
        import types
        from gaudi_amd import sampling_edm as gs
        from gaudi_amd.models_edm import CondPredictor, GaudiModel
        model = GaudiModel.from_engine(eng, eargs)
        cp = CondPredictor.from_engine(model, pargs)
        if closure == "linear":
            def closure_target(_input, _node_mask, _edge_mask, _t):  # generation_guidance.py:200-203, verbatim
                pred = cp(_input, _node_mask, _edge_mask, _t)
                gap = pred[:, 1]
                return -gap
        else:
            def closure_target(_input, _node_mask, _edge_mask, _t):  # a target that is NOT affine in pred (README: "any target")
                import torch as _t_
                pred = cp(_input, _node_mask, _edge_mask, _t)
                return -_t_.tanh(0.05 * pred[:, 1]) * 20.0 + 0.01 * pred[:, 0] ** 2
        cl_args = types.SimpleNamespace(device="cuda", dataset=ds, max_nodes=N)
        nodesxsample = np.full(B, N, np.int64)

    def one_pass(it):
        if closure_target is not None:
            model.sample_offset = rank * B  # same global sample indices every pass (the seed changes)
            model.seed = 1234 + it
            x, h, _, _ = gs.sample_guidance(cl_args, model, closure_target, nodesxsample, scale=0.6, std=1.0)
            x, h, diag = x.numpy(), h.numpy(), model.last_diag
        else:
            x, h, diag = eng.sample(nm, em, seed=1234 + it, sample_offset=rank * B, std=1.0, target_w=tw, scale=0.6)
        t_g = 0.0
        if use_dist:
            tg0 = time.perf_counter()
            x, h = gdist.gather_to_all(x, h, B * world, N, F, device=dev if backend == "nccl" else None)
            t_g = time.perf_counter() - tg0
        gather_s[0] += t_g
        return x, h, diag

    gather_s = [0.0]

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(warmup):
        one_pass(-1 - it)
    eng.profile_reset(True)
    gather_s[0] = 0.0
    sync()
    t0 = time.perf_counter()
    for it in range(steps):
        x, h, diag = one_pass(it)
    dt_own = time.perf_counter() - t0  # this rank's own time, before it waits for the others
    sync()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if use_dist:
        tdev = dev if backend == "nccl" else "cpu"
        tt = torch.tensor([dt], device=tdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the curve must be decomposable: every rank's own time per call and its share spent in the gather
        mine = torch.tensor([dt_own / steps * 1e3, gather_s[0] / steps * 1e3], device=tdev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [[float(v) for v in t_.cpu()] for t_ in allr]
    n_launch, kern_ms, steps_done = eng.profile_get()
    clock_mhz = eng.profile_clock_mhz()  # shader clock the chip held during the last profiled launch (0: not available)
    eng.profile_reset(False)
    assert x.shape[0] == B * world and np.isfinite(x).all()
    if rank != 0:
        return None
    value = steps * B * world / dt
    # ---- roofline of the dominant kernel (sampler_kernel_v<..,192,208>: EDM + predictor fwd/bwd + update) against the matrix
    # pipe.  Issued instructions are counted from the kernel's loop structure (flops.step_mfma_counts; the same numbers
    # rocprofv3's SQ_INSTS_VALU_MFMA_MOPS_F32 / 4 and _BF16 / 32 report): fp32 v_mfma_f32_16x16x4_f32 (2048 FLOP, peak
    # 157.3 TFLOP/s) and, on the split-operand edge GEMMs, v_mfma_f32_16x16x32_bf16 (16384 FLOP, peak 2516.6).
    #   achieved = all issued matrix FLOPs / launch time;  frac = time the pipe needs at peak / launch time
    #            = (F32 / 157.3 + BF16 / 2516.6) / t, i.e. the matrix-pipe busy fraction at the nominal clock;
    #   peak = achieved / frac (the blend of the two peaks this instruction mix could reach).
    # `useful_*` = the factorised algorithm on live edges and unpadded features in fp32 FLOPs (what the issued work is
    # worth); `as_written_*` = the reference's dense concat+Linear formulation, a throughput-equivalent only.
    wv = eng.kernel_variant()[1]
    em_mode = eng.edge_math()[1]  # 0 fp32 instructions, 1 split operands (full LDS weight ring), 2 split (half ring)
    variant = "w4" if wv == 4 else ("w8s" if em_mode else "w8")
    v8g = wv == 8 and eng.node_buffers_global()  # molecules beyond the LDS limit on the 8-wave kernels (node buffers in global memory)
    nb_form = eng.node_buffers_form()  # 2: V8G with P / Q in LDS; 3: a wide group on the full ring, one predictor buffer in global memory
    # workgroups of the launches that actually ran (molecules, or the groups the call packed them into): the issued-instruction
    # model below must describe THAT launch, so the host-side plan is only used when it agrees with it
    run_groups, run_slots = eng.last_launch_shape()
    units, ncols = graph_meta(nm, em) if variant == "w4" else graph_meta8(nm, em)
    if variant != "w4" and run_groups != B:
        # packed: small molecules share a workgroup; node slots > N = WIDE groups (two rounds of edge tiles, up to 2 N node slots)
        G_, _, units2, ncols2 = eng.pack_plan(nm, em, node_slots=run_slots if run_slots > N else None)
        if G_ == run_groups:
            units, ncols = units2, ncols2
        else:
            # (a request that gaudi_sample cut into sub-batches: the last launch's shape describes one cut, the plan the whole
            # batch -- the roofline then prices the unpacked launch)
            sys.stderr.write(f"bench.py: the last launch ran {run_groups} workgroups, the host pack plan of the whole batch says {G_}: "
                             "roofline figures use the unpacked launch\n")
    if v8g and run_groups == B:
        # V8G sampling launches (round 6) compact a molecule's nodes to the front slots: the node GEMMs produce as many columns as
        # the molecule USES (live nodes and nodes that touch a live edge), not "1 + the last used index"
        live = nm > 0
        ee = (em.reshape(B, N, N) != 0) & (live[:, :, None] | live[:, None, :])
        ncols = np.minimum(np.asarray(ncols), (live | ee.any(2) | ee.any(1)).sum(1))
    npairs = units
    G = len(ncols)  # workgroups per call: molecules, or groups of molecules when the call packs
    pa = pargs if guided else None
    mct = 3 if v8g else 2  # column tiles per node-GEMM pass (V8G: 33..48 node columns in one pass of three)
    cnt = np.array([flops.step_mfma_counts(npairs[b], int(ncols[b]), eargs, pa, variant, mct) for b in range(G)], dtype=np.float64).sum(0)
    cnt_edm = np.array([flops.step_mfma_counts(npairs[b], int(ncols[b]), eargs, None, variant, mct) for b in range(G)], dtype=np.float64).sum(0)
    equiv_variant = "w8" if variant == "w8s" else variant  # the same work issued as fp32 matrix instructions
    mfma_step = sum(flops.step_mfma_issued(npairs[b], int(ncols[b]), eargs, pa, equiv_variant, mct) for b in range(G))
    edm_only = sum(flops.step_mfma_issued(npairs[b], int(ncols[b]), eargs, None, equiv_variant, mct) for b in range(G))
    useful_step = B * flops.step_flops_useful(live_edges, live_nodes, F, eargs, pa, K)
    written_step = B * flops.step_flops_as_written(N, F, eargs, pa, K)
    # per launch: `steps_done` reverse steps + one decode pass (= one EDM evaluation) per call, over n_launch launches
    per_launch = (cnt * steps_done + cnt_edm * steps) / max(n_launch, 1)
    f32_flop, bf_flop = per_launch[0] * flops.FLOP_MFMA_F32, per_launch[1] * flops.FLOP_MFMA_BF16
    evals = steps_done + steps * (edm_only / max(mfma_step, 1))
    wvar = equiv_variant if variant != "w8s" else "w8s"
    if variant == "w4":
        wstream = flops.step_weight_stream_bytes(eargs, pa, wvar)
    else:  # mean over the workgroups: an edge-level matrix is streamed once per round of eight tiles
        wstream = float(np.mean([flops.step_weight_stream_bytes(eargs, pa, wvar, rounds=max(1, -(-int(u) // 8))) for u in units]))
    avg_launch_ms = kern_ms / max(n_launch, 1)
    t_launch = avg_launch_ms * 1e-3
    achieved = (f32_flop + bf_flop) / t_launch / 1e12
    frac = (f32_flop / flops.PEAK_F32_TFLOPS + bf_flop / flops.PEAK_BF16_TFLOPS) / 1e12 / t_launch
    assert 0.0 < frac <= 1.0, f"roofline.frac = {frac}: the issued-instruction model or the timing is wrong"
    peak = achieved / frac
    equiv_launch = 2048.0 * (mfma_step * steps_done + edm_only * steps) / max(n_launch, 1)
    stash = 4 * pargs["n_layers"] * ((3 * N * 208 + 4 * N) + 4 * 32 * 208 * 2) if guided else 0
    hbm_bytes_launch = flops.step_bytes_fused(B, N, F, 0, stash) * steps_done / max(n_launch, 1) + wbytes
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and B == (1024 if hetero else 256) and a.steps_per_launch == 25 and T == 1000:  # shape of the PMC passes
        try:
            rec = json.load(open(pmc))
            # hardware counters are collected in separate rocprofv3 passes (tools/profile_round.sh) and stored; they describe the
            # kernels of the tree they were measured on -- any edit of the kernel sources since then makes the figure null here
            from gaudi_amd import build as _build
            traffic = rec.get(f"{workload}_bytes_per_launch") if rec.get("csrc_sha256") == _build.csrc_digest() else None
        except Exception:
            traffic = None
    label = {"c3": f"C3: cc-PBH {N}-ring, batch={B}/GPU, {T} steps, HOMO-LUMO-gap guidance (scale 0.6)",
             "c2": f"C2: cc-PBH {N}-ring, batch={B}/GPU, {T} steps, unconditional EDM",
             "c4": f"C4: PASs-like hetero, 3-10 rings (6-20 graph nodes, N=20: the reference's own cap, "
                   f"data/aromatic_dataloader.py:285), batch={B}/GPU, {T} steps, multi-objective (OPV) guidance",
             "c4x": f"C4 read literally: PASs-like hetero, 6-20 rings (12-40 graph nodes, N=40; beyond the reference dataset's cap), "
                    f"batch={B}/GPU, {T} steps, multi-objective (OPV) guidance"}[workload]
    if world > 1 and workload == "c3":
        label = f"C5: {B * world} guided cc-PBH 11-ring samples sharded over {world} GPUs ({B}/GPU), {T} steps, one RCCL all_gather at the end"
    return {
        "metric": ("guided" if guided else "unguided") + f" molecules/sec ({T}-step)",
        "value": value, "unit": "molecules/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "edge_gemm_math": "fp16 hi/lo pairs of both operands, 3 piece products, f32 accumulate" if variant == "w8s" else "f32",
        "node_gemm_math": "fp16 hi/lo pairs of both operands, f32 accumulate" if variant == "w8s" else "f32",
        "data": "synthetic (seeded default-init weights, on-device Philox noise)",
        "config": {"workload": label + (f"; reference-form closure ({closure}) through gaudi_amd.sampling_edm.sample_guidance"
                                        if closure else ""), "global_batch": B * world, "n_nodes": N, "diffusion_steps": T,
                   "workgroups_per_call": G, "node_slots_per_workgroup": run_slots,
                   "edm": "nf=192,n_layers=9", "predictor": "nf=196,n_layers=12" if guided else None,
                   "parallelism": f"sample-sharded x{world}, one RCCL all_gather per call",
                   "steps_per_launch": a.steps_per_launch,
                   "max_graph_nodes": "resident kernels: 22 at these hidden sizes (one molecule's node buffers in 160 KiB of LDS); "
                                      "beyond that node buffers in global memory: the V8G kernels (8 waves) while no node has more "
                                      "than 32 live edges, else the V4G kernels (4 waves): checked at N = 40, bounded by the edge "
                                      "lists in LDS (a complete graph of about 60 nodes)"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": frac, "traffic": traffic,
                     # the peaks above are priced at the nominal 2.4 GHz; under this kernel's load the chip holds less (workgroup 0
                     # reads the shader-clock and the constant 100 MHz counters at kernel entry and exit: gaudi_profile_clock)
                     "clock_mhz": clock_mhz or None, "nominal_clock_mhz": NOMINAL_CLOCK_MHZ,
                     "frac_at_clock": (frac * NOMINAL_CLOCK_MHZ / clock_mhz) if clock_mhz else None,
                     "parity_rel_err": gate_out["rel_err"] if gate_out is not None else None,
                     "kernel": "sampler_kernel_v<V8T<1,true,%d>,192,%s> (V8G: 8 waves, node buffers in a per-workgroup global scratch%s, several rounds of edge tiles)" % (nb_form, "208" if guided else "0", " except P and Q" if nb_form == 2 else "") if v8g else
                               "sampler_kernel_v<V8T<1,true,0,false,true>,192,208> (wide groups on the full ring: several rounds of edge tiles, one of the predictor's five node buffers in a per-workgroup global scratch)" if nb_form == 3 else
                               ("sampler_kernel_v<%s%s,192,%s>" % ({"w4": "V4", "w8": "V8", "w8s": "V8H" if em_mode == 2 else "V8S"}[variant],
                                                                    " (MR: several rounds of edge tiles)" if guided and variant != "w4" and int(np.max(units)) > 8 else
                                                                    " (FR instantiation: more than 16 node slots, kern8s2_*.hip)" if variant == "w8s" and em_mode == 1 and run_slots > 16 else "",
                                                                    "208" if guided else "0")) if not (variant == "w4" and N > 22) else
                               "sampler_kernel_g<V4G,192,0> + sampler_kernel_g<V4G,0,208> (node buffers in global memory; two launches "
                               "per guided step, averaged together)",
                     "kernel_variant": {"w4": "4 waves per molecule, fp32 matrix instructions",
                                        "w8": "8 waves per molecule (two per SIMD), fp32 matrix instructions",
                                        "w8s": "8 waves per molecule (two per SIMD); edge GEMMs (round 5): both operands as fp16 pairs "
                                               "hi + lo (22 significant bits, power-of-two scales per network / per edge column), 3 piece "
                                               "products accumulated in fp32 on the fp16 matrix pipe (error vs float64 at the level of the "
                                               "fp32 instruction's own: tests/test_gpu_split.py); "
                                               "node GEMMs (round 5): both operands as fp16 pairs hi + 2^-11 lo (22 significant bits, "
                                               "power-of-two scales per network / per node), 3 piece products accumulated in fp32 "
                                               "on the fp16 matrix pipe, the same bytes per weight as fp32 (closer to float64 than "
                                               "the fp32 instruction: profiles/r05b_node_gemm_h_microbench.txt); "
                                               "activation stash written / read non-temporally"}[variant],
                     "flops_basis": "issued matrix instructions counted from the kernel's loop structure (padding included): "
                                    "v_mfma_f32_16x16x4_f32 equivalents (a 4x4x1_16B instruction = 1/4) x 2048 FLOP at 157.3 TFLOP/s + "
                                    "(v_mfma_f32_16x16x32_bf16 + v_mfma_f32_16x16x32_f16) x 16384 FLOP at 2516.6 TFLOP/s; frac = matrix-pipe "
                                    "time at peak / launch time; = SQ_INSTS_VALU_MFMA_MOPS_F32 / 4 and (_BF16 + _F16) / 32 in profiles/",
                     "l2_weight_stream": {
                         "bytes_per_workgroup_step": wstream,
                         "achieved_TBps": G * wstream * evals / max(n_launch, 1) / t_launch / 1e12,
                         "note": "packed weights every workgroup streams from L2 per network evaluation (no reuse across workgroups "
                                 "of a CU: one molecule or packed group per workgroup); MI355X_MICROARCH.md: 16.8-18.8 TB/s chip-wide "
                                 "for rows gathered from L2 hits, 34.5 TB/s L2 peak -- the second ceiling of this design beside the matrix pipe"},
                     "issued_fp32_mfma_per_launch": per_launch[0], "issued_bf16_mfma_per_launch": per_launch[1],
                     "peak_fp32_matrix_tflops": flops.PEAK_F32_TFLOPS, "peak_bf16_matrix_tflops": flops.PEAK_BF16_TFLOPS,
                     "fp32_equivalent_tflops": equiv_launch / t_launch / 1e12,
                     "fp32_equivalent_frac_of_fp32_peak": equiv_launch / t_launch / 1e12 / flops.PEAK_F32_TFLOPS,
                     "issued_gflop_per_molecule_step": 2048.0 * mfma_step / B / 1e9,
                     "useful_gflop_per_molecule_step": useful_step / B / 1e9,
                     "as_written_gflop_per_molecule_step": written_step / B / 1e9,
                     "useful_tflops": useful_step * evals / (kern_ms * 1e-3) / 1e12,
                     "useful_frac": useful_step * evals / (kern_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,  # of the fp32 peak
                     "as_written_tflops_equivalent": written_step * evals / (kern_ms * 1e-3) / 1e12,
                     "live_edges_per_molecule": live_edges,
                     "avg_launch_ms": avg_launch_ms, "launches": n_launch,
                     "hbm_algorithmic_gbps": hbm_bytes_launch / (avg_launch_ms * 1e-3) / 1e9,
                     "hbm_frac": hbm_bytes_launch / (avg_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS},
        "diag": diag,
        **({"parity_gate": gate_out} if gate_out is not None else {}),
        **({"rccl_ranks": world, "per_rank_ms": [r[0] for r in per_rank_ms], "gather_ms": [r[1] for r in per_rank_ms],
            "per_rank_note": "per call: each rank's own wall time before the closing barrier, and the part of it spent in the "
                             "all_gather (host staging + collective)"} if per_rank_ms is not None else {}),
    }


def main():
    a = parse()
    # RCCL prints a version banner on stdout when a communicator is created; the contract is ONE JSON line on stdout.
    # Everything this process (and the libraries it loads) writes to fd 1 goes to stderr; the JSON line is written to the
    # saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    if a.workload == "stability":
        return bench_stability(a, emit)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        # convenience: relaunch under torch.distributed.run as a CHILD process (never exec after GPU init)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, stdout=json_fd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    from gaudi_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sampler has no CPU fallback")
    # GAUDI_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing test on a 1-GPU box); the real
    # multi-GPU run uses RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("GAUDI_BENCH_BACKEND", "nccl")
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    use_dist = world > 1 or a.dist
    if use_dist:
        if world == 1:  # --dist without a launcher: a one-rank group on this GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29534")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    T, K = a.diffusion_steps, 5
    # N = 1: BASELINE configs[2] (C3, 256 molecules) -- or configs[1] / configs[3] with --workload.  N > 1: configs[4]
    # (C5): 1024 guided samples per GPU, 8192 over 8 GPUs.
    B = a.batch or (1024 if (a.workload in ("c4", "c4x") or world > 1) else 256)
    engines = {}
    out = run_workload(a, engines, a.workload, B, a.steps, a.warmup, rank, world, dev, backend, T, use_dist=use_dist,
                       gate=not a.no_parity_gate, closure=a.closure if a.workload == "c3" and world == 1 else None)
    gate_failed = rank == 0 and "parity_gate" in out and not out["parity_gate"]["passed"]
    if use_dist:  # every rank leaves with the same exit code (the gate runs on rank 0: one CPU step, not one per GPU)
        flag = [gate_failed]
        dist.broadcast_object_list(flag, src=0)
        gate_failed = bool(flag[0])
    if gate_failed:
        if rank == 0:
            sys.stderr.write("bench.py: PARITY GATE FAILED: " + json.dumps(out["parity_gate"]) + "\n")
            emit({"error": "parity gate failed", "parity_gate": out["parity_gate"]})
        os._exit(3)
    if rank == 0:
        if use_dist:
            out["config"]["collective"] = (f"{backend} all_gather per call (world {world}"
                                           + (f", RCCL {'.'.join(map(str, torch.cuda.nccl.version()))}" if backend == "nccl" else "") + ")")
        if world == 1 and not a.no_secondary and a.workload == "c3":
            # the other single-GPU configurations, timed by the same harness (short: one or two calls each)
            sec = {}
            # (every line is gated like the headline: one teacher-forced full-batch step of the engine that is about to be timed
            # against the C++ port, untimed -- round 6: c3_b1024's wide-group launch too, all 1 024 molecules)
            for wl, b, st, wu, math, gt in (("c2", 256, 2, 1, None, True), ("c4", 1024, 2, 1, None, True), ("c3_b1024", 1024, 2, 1, None, True),
                                            ("c3_fp32_mfma", 256, 2, 1, "fp32", True), ("c2_fp32_mfma", 256, 2, 1, "fp32", True),
                                            ("c4x", 1024, 1, 1, None, True)):  # BASELINE config 4 read literally: 12-40 graph nodes (V8G kernels)
                r = run_workload(a, engines, wl.split("_")[0], b, st, wu, rank, world, dev, backend, T, edge_math=math,
                                 gate=gt and not a.no_parity_gate)
                sec[wl] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "steps": st,
                           "warmup": wu, "ms_per_step": r["ms_per_step"], "roofline_frac": r["roofline"]["frac"],
                           "useful_frac": r["roofline"]["useful_frac"],
                           "fp32_equivalent_frac_of_fp32_peak": r["roofline"]["fp32_equivalent_frac_of_fp32_peak"],
                           "avg_launch_ms": r["roofline"]["avg_launch_ms"], "kernel": r["roofline"]["kernel"],
                           "clock_mhz": r["roofline"]["clock_mhz"], "frac_at_clock": r["roofline"]["frac_at_clock"],
                           "edge_gemm_math": r["edge_gemm_math"]}
                if "parity_gate" in r:
                    sec[wl]["parity_gate"] = {k: r["parity_gate"][k] for k in ("rel_err", "tol", "passed")}
                    if not r["parity_gate"]["passed"]:
                        sys.stderr.write(f"bench.py: PARITY GATE FAILED on the secondary line {wl}: " + json.dumps(r["parity_gate"]) + "\n")
                        emit({"error": f"parity gate failed ({wl})", "parity_gate": r["parity_gate"]})
                        os._exit(3)
            # the reference-form call: C3 through sampling_edm.sample_guidance with the closure the
            # reference ships (affine in pred: recognised and run on the fused kernel) and with one that is not (callback path:
            # two launches per step around torch.autograd on the [B,K] leaf)
            for wl, cl, st, wu in (("c3_closure_linear", "linear", 2, 1), ("c3_closure_nonlinear", "nonlinear", 1, 1)):
                try:
                    r = run_workload(a, engines, "c3", 256, st, wu, rank, world, dev, backend, T, closure=cl)
                    sec[wl] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "steps": st,
                               "warmup": wu, "ms_per_step": r["ms_per_step"], "launches": r["roofline"]["launches"],
                               "ratio_to_fused_headline": r["value"] / out["value"]}
                except Exception as exc:  # a secondary line must not cost the headline
                    sec[wl] = {"error": repr(exc)}
            out["secondary"] = sec
        if world == 1 and not a.no_cpu_baseline:
            guided = a.workload in ("c3", "c4", "c4x")
            ref8 = 0.058 if guided else 0.175  # the reference's own PyTorch-CPU path on the 8-core build container
            try:  # the baseline leg must never cost the measured line (no g++ / no AVX2 / a stale build on this host)
                from oracle import build_cpu
                if not build_cpu.cpu_ok():
                    raise RuntimeError("host CPU lacks AVX2/FMA: the C++ port cannot run here")
                out["cpu_baseline"] = cpu_baseline(synth.edm_args(diffusion_steps=T), synth.pred_args(),
                                                    synth.synth_edm_state_dict(synth.edm_args(diffusion_steps=T), 1, seed=0),
                                                    synth.synth_predictor_state_dict(synth.pred_args(), 1, K, seed=1), guided, T,
                                                    256)
                out["speedup_vs_cpu_baseline"] = out["value"] / max(out["cpu_baseline"]["value"],
                                                                     out["cpu_baseline"]["reference_torch_cpu_8core"])
            except Exception as exc:
                out["cpu_baseline"] = {"error": repr(exc), "reference_torch_cpu_8core": ref8, "unit": "molecules/s",
                                       "kind": "reference", "cores": 8,
                                       "sample": "C++ port unavailable on this host; the figure is the reference's PyTorch-CPU "
                                                 "path measured on the 8-core build container (BASELINE.md section 2)", "value": ref8}
                out["speedup_vs_cpu_baseline"] = out["value"] / ref8
        emit(out)
    for eng, *_ in engines.values():
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
