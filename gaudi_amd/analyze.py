"""Host mirror of analyze/analyze.py (graph-of-rings stability) on top of gaudi_check_stability.

Same call shapes as the reference -- ``check_stability(positions, ring_type, tol, dataset)`` for one molecule,
``analyze_validity_for_molecules(molecule_list, tol, dataset)`` for a list of ``(positions, ring_type)`` pairs,
``positions2adj(x, ring_type, tol, dataset)`` for a batch -- but all molecules of a call are checked in ONE kernel launch
(one wavefront per molecule).  The geometry tables (ring-ring distance windows, 3-ring angle windows, dihedral
thresholds: utils/helpers.py:11-157) ship as data in ``gaudi_amd/data/ring_tables.json``.  There is no CPU
implementation behind these functions: without the HIP library they raise GaudiError."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from ._lib import GaudiError, RingTables, StabilityAux, f32, fptr

FLAG_NAMES = ("orientation_nodes", "dist_stable", "connected", "angels3", "angels4")
_TABLES = None


def ring_tables() -> dict:
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ring_tables.json")) as f:
            _TABLES = json.load(f)
    return _TABLES


def rings_list(dataset: str):
    """RINGS_LIST[dataset] (data/aromatic_dataloader.py:31-35)."""
    return ring_tables()["rings"][_key(dataset)]


def _key(dataset: str) -> str:
    if dataset == "peri":  # same rings and distances as cata (utils/helpers.py:151-155); angle tables are cata's
        return "cata"
    if dataset not in ("cata", "hetro"):
        raise GaudiError(f"no geometry tables for dataset {dataset!r}")
    return dataset


def c_tables(dataset: str, tol: float = 0.1) -> RingTables:
    """gaudi_ring_tables for one dataset."""
    ds = _key(dataset)
    T = ring_tables()
    R = len(T["rings"][ds])
    t = RingTables()
    t.n_types = R
    t.orientation = int(dataset != "cata")  # analyze/analyze.py:66
    t.check_dihedrals = int(dataset != "hetro")  # analyze/analyze.py:40
    t.tol = float(tol)
    t.min_dist = float(T["min_dist"][ds])
    for i in range(R):
        for j in range(R):
            t.dist_lo[i][j] = T["dist_lo"][ds][i][j]
            t.dist_hi[i][j] = T["dist_hi"][ds][i][j]
        wins = T["a3"][ds][i]
        t.a3_count[i] = len(wins)
        for q, (lo, hi) in enumerate(wins):
            t.a3_lo[i][q], t.a3_hi[i][q] = lo, hi
    t.a4_0, t.a4_180 = T["a4"][ds]["0"], T["a4"][ds]["180"]
    return t


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def _engine(engine):
    if engine is not None:
        return engine
    from .engine import Engine
    return Engine.default()


def check_stability_batch(x, ring_type, n_nodes, tol=0.1, dataset="cata", engine=None, want_adj=False, want_aux=False):
    """x [B,N,3], ring_type [B,N] (int) and n_nodes [B] with each molecule's valid nodes first
    -> flags [B,5] bool (FLAG_NAMES order) (+ dist, adj [B,N,N]) (+ aux record array)."""
    eng = _engine(engine)
    x = f32(x)
    B, N = x.shape[0], x.shape[1]
    ty = np.ascontiguousarray(_np(ring_type), dtype=np.int32).reshape(B, N)
    nn = np.ascontiguousarray(_np(n_nodes), dtype=np.int32).reshape(B)
    flags = np.zeros((B, 5), np.uint8)
    dist = np.zeros((B, N, N), np.float32) if want_adj else None
    adj = np.zeros((B, N, N), np.float32) if want_adj else None
    aux = (StabilityAux * B)() if want_aux else None
    t = c_tables(dataset, tol)
    rc = eng.lib.gaudi_check_stability(eng.h, C.byref(t), B, N, fptr(x), ty.ctypes.data_as(C.POINTER(C.c_int32)),
                                       nn.ctypes.data_as(C.POINTER(C.c_int32)),
                                       flags.ctypes.data_as(C.POINTER(C.c_uint8)), fptr(dist), fptr(adj), aux)
    eng._check(rc, "gaudi_check_stability")
    out = [flags.astype(bool)]
    if want_adj:
        out += [dist, adj]
    if want_aux:
        out.append(np.array([tuple(getattr(a, f) for f, _ in StabilityAux._fields_) for a in aux],
                            dtype=[(f, np.int32 if "n_" in f else np.float32) for f, _ in StabilityAux._fields_]))
    return out[0] if len(out) == 1 else tuple(out)


def _pack(molecule_list):
    B = len(molecule_list)
    if B == 0:
        raise GaudiError("empty molecule list")
    xs = [_np(x).astype(np.float32).reshape(-1, 3) for x, _ in molecule_list]
    ts = []
    for _, t in molecule_list:
        t = _np(t)
        ts.append((t.argmax(1) if t.ndim == 2 else t).astype(np.int32))  # analyze/analyze.py:62-63
    N = max(1, max(len(x) for x in xs))
    X = np.zeros((B, N, 3), np.float32)
    T = np.zeros((B, N), np.int32)
    nn = np.zeros(B, np.int32)
    for b, (x, t) in enumerate(zip(xs, ts)):
        if len(x) != len(t):
            raise GaudiError(f"molecule {b}: {len(x)} positions but {len(t)} ring types")
        X[b, : len(x)], T[b, : len(x)], nn[b] = x, t, len(x)
    return X, T, nn


def check_stability(positions, ring_type, tol=0.1, dataset="cata", engine=None) -> dict:
    """analyze/analyze.py:50-100 for one molecule -> {orientation_nodes, dist_stable, connected, angels3, angels4}."""
    X, T, nn = _pack([(positions, ring_type)])
    flags = check_stability_batch(X, T, nn, tol, dataset, engine)
    return {k: bool(v) for k, v in zip(FLAG_NAMES, flags[0])}


def analyze_validity_for_molecules(molecule_list, tol=0.1, dataset="cata", engine=None):
    """analyze/analyze.py:139-177 -> (validity_dict, molecule_stable_list); one launch for the whole list."""
    X, T, nn = _pack(molecule_list)
    flags = check_stability_batch(X, T, nn, tol, dataset, engine)
    stable = flags.all(1)
    n = float(len(molecule_list))
    d = {"mol_stable": int(stable.sum()) / n}
    for i, k in enumerate(FLAG_NAMES):
        d[k] = int(flags[:, i].sum()) / n
    d["molecule_stable_bool"] = [bool(s) for s in stable]
    return d, [m for m, s in zip(molecule_list, stable) if s]


def positions2adj(x, ring_type, tol=0.1, dataset="cata", engine=None):
    """utils/helpers.py:167-190: x [B,n,3], ring_type [B,n] or one-hot [B,n,R] -> (dist, adj) [B,n,n].  Every node is
    treated as a ring (the reference function knows nothing about orientation nodes)."""
    x = f32(_np(x))
    t = _np(ring_type)
    if t.ndim == 3:
        t = t.argmax(2)
    B, N = x.shape[0], x.shape[1]
    tb = c_tables(dataset, tol)
    tb.orientation = 0
    eng = _engine(engine)
    flags = np.zeros((B, 5), np.uint8)
    dist = np.zeros((B, N, N), np.float32)
    adj = np.zeros((B, N, N), np.float32)
    ty = np.ascontiguousarray(t, dtype=np.int32)
    nn = np.full(B, N, np.int32)
    rc = eng.lib.gaudi_check_stability(eng.h, C.byref(tb), B, N, fptr(x), ty.ctypes.data_as(C.POINTER(C.c_int32)),
                                       nn.ctypes.data_as(C.POINTER(C.c_int32)),
                                       flags.ctypes.data_as(C.POINTER(C.c_uint8)), fptr(dist), fptr(adj), None)
    eng._check(rc, "gaudi_check_stability")
    return dist, adj
