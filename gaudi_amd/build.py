"""Build libgaudi_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The kernel instantiations are spread over kern_*.hip translation units which are compiled in
parallel and linked with the host TU (gaudi_hip.hip)."""
from __future__ import annotations

import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libgaudi_hip.so")
HEADERS = ["device_common.h", "edm_device.h", "pred_device.h", "sampler_kernel.h", "w8_common.h", "w8_split.h", "w8_nodes_f16.h", "w8_edm.h", "w8_pred.h", "pred_host.inc", "stability.inc",
           os.path.join("..", "..", "include", "gaudi_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]


def _sources():
    return ["gaudi_hip.hip"] + sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "kern*_*.hip")))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def csrc_digest() -> str:
    """sha256 over the kernel sources (csrc/*.h, *.inc, *.hip): what a stored hardware-counter figure (profiles/pmc_traffic.json)
    is valid for.  Source text, not the built library: two machines build different bytes from the same tree."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.hip"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build() -> bool:
    deps = [os.path.join(CSRC, f) for f in _sources() + HEADERS]
    return _stale(LIB, deps)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    todo = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdrs):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=CSRC, check=True)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(cc, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
