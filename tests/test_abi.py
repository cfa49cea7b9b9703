"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/commu_hip.h declares, and the ctypes prototype table covers exactly those symbols.
No compute call is made (there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "commu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(commu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from commu_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        import sys
        subprocess.check_call([sys.executable, os.path.join(ROOT, "commu-code_amd", "build.py")])
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in commu_hip.h but not exported"
    assert set(_lib.PROTOTYPES) == set(syms)
    assert b"gfx950" in lib.commu_hip_version()


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU: wrappers refuse CPU tensors."""
    import torch
    from commu_amd import ops
    from commu_amd._lib import CommuHipError
    a = torch.zeros(4, 32, dtype=torch.bfloat16)
    with pytest.raises(CommuHipError):
        ops.gemm_nt(a, a)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "commu-code_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle"
                assert "/root/reference" not in text
