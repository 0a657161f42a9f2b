"""CPU checks of the drop-in boundary: libefx_hip.so loads, exports every symbol include/efx.h declares, and
fails loudly (no CPU fallback) when no HIP device is present.  No compute calls."""
import ctypes
import os
import re

import pytest

import cef_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cef():
    import __graft_entry__
    __graft_entry__.build()
    return cef_loader.load()


def test_header_symbols_are_exported(cef):
    hdr = open(os.path.join(ROOT, "include", "efx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(efx_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = cef.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"not exported: {missing}"
    assert sorted(declared) == sorted(cef.ABI_SYMBOLS)
    assert lib.efx_version() == 100


def test_default_params(cef):
    p = cef.Params()
    cef.lib().efx_default_params(ctypes.byref(p))
    # EfficientFeatures::create defaults, cuda_efficient_features.h:47-48
    assert (p.nfeatures, p.nlevels, p.first_level, p.fast_threshold, p.nonmax_radius, p.descriptor_type) == (5000, 8, 0, 20, 15, 2)
    assert abs(p.scale_factor - 1.2) < 1e-7


def test_no_cpu_fallback(cef):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cef.EfxError) as e:
        cef.EfficientFeatures.create()
    assert e.value.status == -4
    with pytest.raises(cef.EfxError):
        cef.BAD.create(1.0)


def test_product_does_not_reference_oracle():
    """The product path must never import, link or call the oracle."""
    pkg = os.path.join(ROOT, "cuda-efficient-features_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".S")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "efxo_" not in text and "pyoracle" not in text and "efx_oracle" not in text, f
