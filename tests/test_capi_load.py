"""CPU-side checks of the drop-in boundary: the C-ABI library builds with hipcc for gfx950,
loads, and exports every symbol include/vpin_hip.h declares.  No compute calls (no GPU here)."""
import os

import pytest

import vpin_amd
from vpin_amd import build as vbuild


@pytest.fixture(scope="module")
def built():
    return vbuild.build()


def test_library_builds_and_loads(built):
    assert os.path.exists(built)
    L = vpin_amd.lib()
    assert L.vpin_abi_version() >= 1


def test_every_declared_symbol_is_exported(built):
    declared = vpin_amd.declared_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if s not in vpin_amd.exported_symbols()]
    assert not missing, f"declared in include/vpin_hip.h but not exported: {missing}"


def test_no_device_is_a_loud_error(built):
    """Without a GPU the product must fail, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vpin_amd.VpinError) as ei:
        vpin_amd.Context(0)
    assert ei.value.code == -2  # VPIN_ENODEV


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under vpin_amd/ or include/ may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base in ("vpin_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(root, base)):
            if os.sep + "lib" in dp:
                continue
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    with open(os.path.join(dp, fn), errors="ignore") as f:
                        txt = f.read()
                    if "oracle/" in txt or "oracle_lib" in txt or "liboracle" in txt:
                        if fn == "__init__.py" and base == "vpin_amd":
                            continue
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
