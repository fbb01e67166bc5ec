"""CPU-only: the C-ABI library builds for gfx950, loads, and exports exactly what include/*.h declares."""
import ctypes
import os
import re

from conftest import ROOT


HEADERS = ("cpmrcnn_hip.h",)      # the boundary


def declared_symbols(headers=HEADERS):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(cpm_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (brings the HIP runtime the library binds to)
    from pet.lib.ops import _hip
    L = _hip.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "include/cpmrcnn_hip.h declares %s but the library does not export it" % n
    assert L.cpm_abi_version() == 1


def test_no_undeclared_exports():
    import subprocess
    from pet.lib.ops import _hip
    out = subprocess.check_output(["nm", "-D", "--defined-only", _hip.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T cpm_" in l)
    assert exported == declared_symbols()


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly without a GPU tensor."""
    import pytest
    import torch
    from pet.lib.ops import _C
    with pytest.raises(RuntimeError):
        _C.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)
    with pytest.raises(RuntimeError):
        _C.roi_align_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 0.25, 7, 7, 2, False, 0)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cpm-r-cnn_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), "%s mentions the oracle" % f
