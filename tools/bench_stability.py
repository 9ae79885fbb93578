#!/usr/bin/env python3
"""Throughput of the graph-of-rings stability kernel (gaudi_check_stability) on replicated golden molecules.

    python tools/bench_stability.py [--molecules 262144] [--dataset cata] > profiles/<round>_stability.json

Prints one JSON line: whole-call rate (H2D + kernel + D2H + host packing excluded), kernel-only rate from HIP events on
the handle's stream, algorithmic bytes per molecule, and the numpy oracle's rate on a bounded sample as the CPU baseline."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=262144)
    ap.add_argument("--dataset", default="cata")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=512)
    a = ap.parse_args()
    from gaudi_amd import analyze
    from gaudi_amd.engine import Engine
    g = np.load(os.path.join(ROOT, "tests", "golden", "g11_stability.npz"))
    ds = a.dataset
    X0, T0, n0 = g[f"{ds}_x"], g[f"{ds}_types"], g[f"{ds}_n"]
    idx = np.arange(a.molecules) % len(n0)
    X = np.ascontiguousarray(X0[idx])
    T = np.ascontiguousarray(np.maximum(T0[idx], 0).astype(np.int32))
    nn = np.ascontiguousarray(n0[idx].astype(np.int32))
    eng = Engine(0)
    flags = analyze.check_stability_batch(X, T, nn, 0.1, ds, engine=eng)  # warm-up + correctness vs the fixture
    assert np.array_equal(flags, g[f"{ds}_flags"].astype(bool)[idx])
    eng.profile_reset(True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        analyze.check_stability_batch(X, T, nn, 0.1, ds, engine=eng)
    wall = (time.perf_counter() - t0) / a.reps
    n_launch, kernel_ms = eng.stability_profile_get()
    kernel_s = kernel_ms / 1e3 / max(n_launch, 1)
    bytes_per_mol = X.shape[1] * 3 * 4 + X.shape[1] * 4 + 4 + 5
    from oracle import stability_oracle as S
    m = min(a.cpu_sample, a.molecules)
    t0 = time.perf_counter()
    for i in range(m):
        S.check_stability(X[i, : nn[i]], T[i, : nn[i]], dataset=ds)
    cpu = m / (time.perf_counter() - t0)
    print(json.dumps(dict(metric="stability-checked molecules/sec", dataset=ds, molecules=a.molecules, N=int(X.shape[1]),
                          whole_call_mol_per_s=a.molecules / wall, kernel_mol_per_s=a.molecules / kernel_s,
                          kernel_ms=kernel_s * 1e3, algorithmic_bytes_per_molecule=bytes_per_mol,
                          kernel_algorithmic_GBps=a.molecules * bytes_per_mol / kernel_s / 1e9,
                          cpu_oracle_mol_per_s=cpu, cpu_sample=m, cpu_cores=1, stable_fraction=float(flags.all(1).mean()))))


if __name__ == "__main__":
    main()
