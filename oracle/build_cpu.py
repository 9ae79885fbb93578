"""TEST INFRASTRUCTURE ONLY: build oracle/gaudi_cpu.cpp into oracle/_build/libgaudi_cpu.so (git-ignored, travels to the GPU
box with the snapshot) and bind it with ctypes.  Used by tests/test_cpu_port.py and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "gaudi_cpu.cpp")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libgaudi_cpu.so")
FP = C.POINTER(C.c_float)


def cpu_ok() -> bool:
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " avx2" in flags and " fma" in flags


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fopenmp", "-fno-math-errno", "-shared", "-fPIC", SRC,
                    "-o", LIB], check=True)
    return LIB


class EdmCfg(C.Structure):
    _fields_ = [("F", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("S", C.c_int32), ("attention", C.c_int32),
                ("use_tanh", C.c_int32), ("coords_range", C.c_float), ("norm_constant", C.c_float), ("normf", C.c_float)]


class PredCfg(C.Structure):
    _fields_ = [("F", C.c_int32), ("K", C.c_int32), ("H", C.c_int32), ("L", C.c_int32), ("attention", C.c_int32),
                ("use_tanh", C.c_int32), ("coords_range", C.c_float)]


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, np.float32))


def _p(a):
    return a.ctypes.data_as(FP)


class CpuPort:
    """The C++/OpenMP restatement: phi, predictor (+ gradient), reverse steps.  Same call shapes as oracle/gaudi_oracle.py."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.gcpu_create.restype = C.c_void_p
        L.gcpu_destroy.argtypes = [C.c_void_p]
        L.gcpu_isa.restype = C.c_char_p
        for name in ("gcpu_load_edm", "gcpu_load_pred"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(FP), C.POINTER(C.c_int64)]
        L.gcpu_phi.argtypes = [C.c_void_p, C.c_int, C.c_int, FP, FP, FP, FP, FP]
        L.gcpu_predictor.argtypes = [C.c_void_p, C.c_int, C.c_int, FP, FP, FP, FP, FP, FP, FP]
        L.gcpu_step.argtypes = [C.c_void_p, C.c_int, C.c_int, FP, C.c_float, FP, FP, FP, FP, FP, C.c_float, FP]
        self.h = C.c_void_p(L.gcpu_create())
        self.F = self.K = None

    def close(self):
        if self.h:
            self.lib.gcpu_destroy(self.h)
            self.h = None

    @property
    def isa(self):
        return self.lib.gcpu_isa().decode()

    @property
    def threads(self):
        return int(self.lib.gcpu_threads())

    def set_threads(self, n: int):
        """OpenMP threads of the following calls (every thread owns groups of molecules)."""
        self.lib.gcpu_set_threads(int(n))

    def set_group(self, g: int):
        """Upper bound of the molecules one thread runs together, layer by layer (a weight matrix is read once per group);
        1 = one molecule at a time, the round-3 arrangement.  Results do not depend on it."""
        self.lib.gcpu_set_group(int(g))

    def group_for(self, B: int) -> int:
        return int(self.lib.gcpu_group(int(B)))

    def _load(self, fn, cfg, sd, prefix):
        sd = {k: _f32(v) for k, v in sd.items() if k.startswith(prefix)}
        names = list(sd)
        n = len(names)
        rc = fn(self.h, C.byref(cfg), n, (C.c_char_p * n)(*[k.encode() for k in names]), (FP * n)(*[_p(sd[k]) for k in names]),
                (C.c_int64 * n)(*[sd[k].size for k in names]))
        if rc != 0:
            raise RuntimeError(f"{fn.__name__} failed ({rc})")

    def load_edm(self, args, sd):
        if args.get("aggregation_method", "sum") != "sum":
            raise ValueError("the C++ port restates aggregation_method='sum' (the numpy oracle covers 'mean')")
        if args.get("sin_embedding", False):
            raise ValueError("the C++ port restates sin_embedding=False (the numpy oracle covers the sinusoid edge features)")
        F = np.asarray(sd["dynamics.egnn.embedding.weight"]).shape[1] - 1
        cfg = EdmCfg(F, int(args["nf"]), int(args["n_layers"]), int(args.get("inv_sublayers", 1)), int(bool(args["attention"])),
                     int(bool(args["tanh"])), float(args["coords_range"]), float(args["norm_constant"]),
                     float(args.get("normalization_factor", 1)))
        self._load(self.lib.gcpu_load_edm, cfg, sd, "dynamics.")
        self.F = F

    def load_predictor(self, args, sd):
        F = np.asarray(sd["egnn.embedding.weight"]).shape[1] - 1
        K = np.asarray(sd["egnn.embedding_out.weight"]).shape[0]
        cfg = PredCfg(F, K, int(args["nf"]), int(args["n_layers"]), int(bool(args["attention"])), int(bool(args["tanh"])),
                      float(args["coords_range"]))
        self._load(self.lib.gcpu_load_pred, cfg, sd, "egnn.")
        self.K = K

    @staticmethod
    def _masks(nm, em, B, N):
        return _f32(nm).reshape(B, N), _f32(em).reshape(B, N, N)

    def phi(self, z, t, node_mask, edge_mask):
        z = _f32(z)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        t = _f32(np.broadcast_to(np.asarray(t, np.float32).reshape(-1), (B,)))
        out = np.empty_like(z)
        assert self.lib.gcpu_phi(self.h, B, N, _p(z), _p(t), _p(nm), _p(em), _p(out)) == 0
        return out

    def predictor(self, z, t, node_mask, edge_mask, dpred=None):
        z = _f32(z)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        t = _f32(np.broadcast_to(np.asarray(t, np.float32).reshape(-1), (B,)))
        pred = np.empty((B, self.K), np.float32)
        if dpred is None:
            assert self.lib.gcpu_predictor(self.h, B, N, _p(z), _p(t), _p(nm), _p(em), None, _p(pred), None) == 0
            return pred
        dp = _f32(np.broadcast_to(np.asarray(dpred, np.float32), (B, self.K)))
        grad = np.empty_like(z)
        assert self.lib.gcpu_predictor(self.h, B, N, _p(z), _p(t), _p(nm), _p(em), _p(dp), _p(pred), _p(grad)) == 0
        return pred, grad

    def step(self, coef, t_val, z_t, node_mask, edge_mask, eps_raw, target_w=None, scale=1.0):
        """coef = oracle.step_coefficients(...) dict; target_w None -> unguided."""
        z = _f32(z_t)
        B, N, D = z.shape
        nm, em = self._masks(node_mask, edge_mask, B, N)
        c = _f32([coef["alpha_ts"], coef["eps_coef"], coef["sigma"]])
        eps = _f32(eps_raw)
        tw = None if target_w is None else _f32(target_w)
        out = np.empty_like(z)
        rc = self.lib.gcpu_step(self.h, B, N, _p(c), float(t_val), _p(z), _p(nm), _p(em), _p(eps), None if tw is None else _p(tw),
                                float(scale), _p(out))
        assert rc == 0, rc
        return out


if __name__ == "__main__":
    print(build(force=True))
