"""CPU checks of the C-ABI boundary: the library loads, exports every symbol that
include/gpx.h declares, and fails loudly (no CPU fallback) without a GPU."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    import glob

    names = set()
    for path in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(gpx_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_header_symbols_exported():
    from gpax_amd import _lib

    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.gpx_abi_version() == 2


def test_struct_layout_matches_header():
    import ctypes as C

    from gpax_amd import _lib

    # 6 x int32 (24 B) + 2 x double + 3 pointers = 64 bytes, no padding surprises
    assert C.sizeof(_lib.KernelParams) == 64
    assert _lib.KernelParams.k_scale.offset == 24 and _lib.KernelParams.k_length.offset == 40


def test_no_cpu_fallback_without_gpu():
    """On a machine without a GPU the product path must raise, not compute."""
    import gpax_amd

    try:
        import torch

        has_gpu = torch.cuda.device_count() > 0
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(gpax_amd.GpxError):
        gpax_amd.Core(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gpax_amd/ (nor bench.py's timed
    path) may import it."""
    pkg = os.path.join(ROOT, "gpax_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "gp_oracle" not in src, f
