"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol declared in
include/gaudi_hip.h; no compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gaudi_amd import _lib, build
    build.build()
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, "include", "gaudi_hip.h")).read()
    declared = set(re.findall(r"\b(gaudi_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gaudi_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gaudi_amd._lib import GaudiError
    from gaudi_amd.engine import Engine
    with pytest.raises(GaudiError):
        Engine(0)
