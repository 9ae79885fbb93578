"""ctypes binding of libgaudi_hip.so (include/gaudi_hip.h).  There is NO CPU fallback: if the HIP
library is missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAUDI_LIB", os.path.join(_PKG, "libgaudi_hip.so"))  # GAUDI_LIB: diagnostic builds


class GaudiError(RuntimeError):
    pass


class EdmConfig(C.Structure):
    _fields_ = [("in_node_nf", C.c_int32), ("hidden_nf", C.c_int32), ("n_layers", C.c_int32),
                ("inv_sublayers", C.c_int32), ("attention", C.c_int32), ("tanh", C.c_int32),
                ("coords_range", C.c_float), ("norm_constant", C.c_float), ("normalization_factor", C.c_float),
                ("diffusion_steps", C.c_int32), ("noise_power", C.c_float), ("noise_precision", C.c_float),
                ("norm_values", C.c_float * 3), ("sin_embedding", C.c_int32)]


class PredConfig(C.Structure):
    _fields_ = [("in_nf", C.c_int32), ("out_nf", C.c_int32), ("hidden_nf", C.c_int32), ("n_layers", C.c_int32),
                ("attention", C.c_int32), ("tanh", C.c_int32), ("coords_range", C.c_float)]


class Diag(C.Structure):
    _fields_ = [("max_masked_leak", C.c_float), ("max_cog_rel", C.c_float), ("max_cog_abs", C.c_float),
                ("nan_count", C.c_int32), ("reprojected", C.c_int32)]


class RingTables(C.Structure):
    _fields_ = [("n_types", C.c_int32), ("orientation", C.c_int32), ("check_dihedrals", C.c_int32), ("tol", C.c_double),
                ("min_dist", C.c_double), ("dist_lo", (C.c_double * 16) * 16), ("dist_hi", (C.c_double * 16) * 16),
                ("a3_count", C.c_int32 * 16), ("a3_lo", (C.c_double * 4) * 16), ("a3_hi", (C.c_double * 4) * 16),
                ("a4_0", C.c_double), ("a4_180", C.c_double)]


class StabilityAux(C.Structure):
    _fields_ = [("n_rings", C.c_int32), ("n_edges", C.c_int32), ("n_triplets", C.c_int32), ("n_nan_angles", C.c_int32),
                ("a3_min", C.c_float), ("a3_max", C.c_float), ("a4_min", C.c_float), ("a4_max", C.c_float)]


FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int32)
# gaudi_target_cb(user, B, K, pred, t, dT_dpred_out)
TARGET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, FP, C.c_float, FP)
# gaudi_target_cbz(user, B, N, D, K, z_s, pred, t, dT_dpred_out, dT_dz_out)
TARGET_CBZ = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, FP, FP, C.c_float, FP, FP)

EXPORTS = {
    "gaudi_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gaudi_destroy": (None, [C.c_void_p]),
    "gaudi_last_error": (C.c_char_p, [C.c_void_p]),
    "gaudi_load_edm": (C.c_int, [C.c_void_p, C.POINTER(EdmConfig), C.c_int, C.POINTER(C.c_char_p), C.POINTER(FP),
                                 C.POINTER(C.c_int64)]),
    "gaudi_load_predictor": (C.c_int, [C.c_void_p, C.POINTER(PredConfig), C.c_int, C.POINTER(C.c_char_p),
                                       C.POINTER(FP), C.POINTER(C.c_int64)]),
    "gaudi_get_gamma": (C.c_int, [C.c_void_p, FP]),
    "gaudi_get_step_coefficients": (C.c_int, [C.c_void_p, FP]),
    "gaudi_phi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, FP, FP, FP]),
    "gaudi_predictor_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, FP, FP, FP]),
    "gaudi_predictor_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, FP, FP, FP, FP, FP]),
    "gaudi_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, FP, FP, FP, FP, FP, C.c_float, FP]),
    "gaudi_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, FP, FP, FP, FP]),
    "gaudi_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, C.c_uint64, C.c_int64, FP, C.c_float, FP,
                               C.c_float, FP, FP, FP, C.POINTER(Diag)]),
    "gaudi_sample_cb": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, C.c_uint64, C.c_int64, FP, C.c_float, TARGET_CB,
                                  C.c_void_p, C.c_float, FP, FP, FP, C.POINTER(Diag)]),
    "gaudi_sample_cbz": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, C.c_uint64, C.c_int64, FP, C.c_float, TARGET_CBZ,
                                   C.c_void_p, C.c_float, FP, FP, FP, C.POINTER(Diag)]),
    "gaudi_sample_chain": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, C.c_uint64, C.c_int64, FP, C.c_float, C.c_int, FP]),
    "gaudi_predict_noised": (C.c_int, [C.c_void_p, C.c_int, C.c_int, FP, FP, IP, FP, FP, C.c_uint64, C.c_int64, FP, FP, FP]),
    "gaudi_check_stability": (C.c_int, [C.c_void_p, C.POINTER(RingTables), C.c_int, C.c_int, FP, IP, IP,
                                        C.POINTER(C.c_uint8), FP, FP, C.POINTER(StabilityAux)]),
    "gaudi_stability_profile_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "gaudi_philox_normal": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, FP]),
    "gaudi_host_schedule": (C.c_int, [C.c_int, C.c_float, C.c_float, FP, FP]),
    "gaudi_host_graph_meta": (C.c_int, [C.c_int, C.c_int, FP, FP, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), FP, C.c_int32,
                                        C.POINTER(C.c_int32)]),
    "gaudi_host_pack_matrix": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, FP, FP]),
    "gaudi_host_pack_matrix_split": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, FP, FP]),
    "gaudi_host_pack_matrix_f16": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, FP, FP]),
    "gaudi_host_weight_scale": (C.c_int, [C.c_int, C.POINTER(FP), IP, IP, IP, FP]),
    "gaudi_host_node_operand_offset": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, IP]),
    "gaudi_profile_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "gaudi_profile_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "gaudi_set_steps_per_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "gaudi_set_readout_nodes": (C.c_int, [C.c_void_p, C.c_int]),
    "gaudi_set_fix_noise": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "gaudi_host_graph_meta8": (C.c_int, [C.c_int, C.c_int, FP, FP, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), FP,
                                         C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_int32, C.POINTER(C.c_int32)]),
    "gaudi_kernel_variant": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gaudi_edge_math": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gaudi_node_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gaudi_last_workgroups": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gaudi_host_pack_plan_wide": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, FP, FP, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gaudi_set_plan_hint": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "gaudi_last_warning": (C.c_char_p, [C.c_void_p]),
    "gaudi_abi_version": (C.c_int, []),
    "gaudi_last_family_split": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gaudi_profile_clock": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "gaudi_last_keep_h": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gaudi_host_pack_plan": (C.c_int, [C.c_int, C.c_int, FP, FP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32)]),
}

ABI_VERSION = 7  # include/gaudi_hip.h: GAUDI_ABI_VERSION
_ROUND6_EXPORTS = ("gaudi_last_warning", "gaudi_abi_version", "gaudi_last_family_split", "gaudi_profile_clock", "gaudi_last_keep_h")  # an older A/B library (GAUDI_LIB) lacks them

_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree library and bind every symbol of include/gaudi_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GaudiError(f"{LIB_PATH} not found: build it with `python -m gaudi_amd.build` "
                         "(there is no CPU fallback for the sampler)")
    lib = C.CDLL(LIB_PATH)
    # the host-side packers (gaudi_host_*) are test / tooling entry points: a diagnostic library named by GAUDI_LIB (an A/B
    # build of another revision) may predate some of them; everything else must be there
    lenient = "GAUDI_LIB" in os.environ
    # ... and one whose exported signatures differ (include/gaudi_hip.h: GAUDI_ABI_VERSION) must not have this file's argument
    # types bound to its host-side entry points at all: round 5 inserted a float into gaudi_host_pack_matrix_split, and a
    # call through the wrong prototype passes that float where the old library expects a pointer
    abi = None
    if hasattr(lib, "gaudi_abi_version"):
        lib.gaudi_abi_version.restype, lib.gaudi_abi_version.argtypes = C.c_int, []
        abi = int(lib.gaudi_abi_version())
    if abi != ABI_VERSION and not lenient:
        raise GaudiError(f"{LIB_PATH} exports ABI version {abi}, this package expects {ABI_VERSION}: rebuild it "
                         "(`python -m gaudi_amd.build --force`)")
    for name, (res, args) in EXPORTS.items():
        if lenient and not hasattr(lib, name) and (name.startswith("gaudi_host_") or name in _ROUND6_EXPORTS):
            continue
        if lenient and abi != ABI_VERSION and name.startswith("gaudi_host_"):
            continue  # left unbound on purpose: a call raises instead of corrupting memory
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def fptr(a: np.ndarray | None):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(FP)


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))
