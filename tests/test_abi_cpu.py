"""CPU-side checks of the product boundary: the C-ABI library builds for gfx950, loads,
and exports every symbol include/tron_hip.h declares; the binding lists them all.
No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, PKG


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tron_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tron_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def native():
    so = os.path.join(PKG, "csrc", "libtron_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["bash", os.path.join(PKG, "csrc", "build.sh")])
    from tron import _native
    return _native


def test_header_symbols_exported_and_bound(native):
    syms = declared_symbols()
    assert len(syms) >= 20 and "tron_step_encode" in syms and "tron_replay_sample" in syms
    lib = C.CDLL(native.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/tron_hip.h but not exported"
        assert s in native.SIGNATURES, f"{s} not bound in tron/_native.py"
    assert sorted(native.SIGNATURES) == syms


def test_abi_version_and_strerror(native):
    L = native.lib()
    assert L.tron_abi_version() == native.ABI_VERSION == 13
    assert L.tron_strerror(0) == b"ok"
    assert b"argument" in L.tron_strerror(-1)


def test_no_cpu_fallback(native):
    """Without a HIP device the product refuses to run instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tron.vec import VecTron, DeviceReplay
    with pytest.raises(native.TronNativeError):
        VecTron(4, 10)
    with pytest.raises(native.TronNativeError):
        DeviceReplay(16, 144)
    with pytest.raises(native.TronNativeError):
        native.ptr(torch.zeros(4))          # host tensors are rejected at the boundary


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)
                assert "tron_oracle" not in text and "libtron_oracle" not in text, os.path.join(dirpath, f)
