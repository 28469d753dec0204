"""CPU: the C-ABI library loads, exports every symbol the header declares, and
fails loudly without a GPU (no CPU fallback anywhere in the product)."""
import ctypes as C
import os
import re

import pytest


def _declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "tensoralloy_amd.h")) as fp:
        text = fp.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ta_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from tensoralloy_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tensoralloy_amd.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_struct_layouts_match_header(lib):
    from tensoralloy_amd import _lib
    assert C.sizeof(_lib.Frame) == 40
    assert C.sizeof(_lib.BatchInfo) == 64 and _lib.BatchInfo.nl_ms.offset == 48
    assert _lib.ModelDesc.rcut.offset == 8 and _lib.ModelDesc.eta.offset == 56
    assert _lib.ModelDesc.eam_params.offset == C.sizeof(_lib.ModelDesc) - 72
    assert _lib.ModelDesc.eps.offset == C.sizeof(_lib.ModelDesc) - 64
    assert _lib.ModelDesc.grap_params.offset == C.sizeof(_lib.ModelDesc) - 48
    assert _lib.ModelDesc.n_eam_nets.offset == C.sizeof(_lib.ModelDesc) - 40
    assert _lib.ModelDesc.eam_table_coef.offset == C.sizeof(_lib.ModelDesc) - 16
    assert _lib.ModelDesc.safe_pow.offset == C.sizeof(_lib.ModelDesc) - 8


def test_no_gpu_means_loud_failure(lib):
    """Without a HIP device the engine must raise, never compute on the CPU."""
    from tensoralloy_amd import Engine
    from tests.helpers import make_nn
    if lib.ta_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        Engine(make_nn(["Ni"], 6.0, False, [8]))


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "tensoralloy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                with open(os.path.join(dirpath, f)) as fp:
                    src = fp.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "sf_oracle" not in src, f
