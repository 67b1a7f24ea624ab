"""CPU: the C-ABI library loads and exports every symbol include/mdqe_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "mdqe_hip.h")


def declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdqe_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from mdqe_cvpr2023_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    h = _lib.load_library()
    names = declared()
    assert "mdqe_msda_forward_f32" in names
    for n in names:
        assert hasattr(h, n), f"{n} declared in mdqe_hip.h but not exported"
    assert h.mdqe_version() >= 100
    # every bound signature is declared in the header (no private ABI)
    for n in _lib.SIGNATURES:
        assert n in names


def test_missing_library_is_loud(tmp_path):
    from mdqe_cvpr2023_amd import _lib
    with pytest.raises(_lib.LibraryMissing):
        _lib.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mdqe_cvpr2023_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert "mdqe_oracle" not in s and "import oracle" not in s and "from oracle" not in s, f


def test_gemm_precision_override_is_per_thread():
    """`ops.gemm_precision(...)` (C ABI mdqe_set_gemm_precision_thread) changes the mode of the CALLING thread's launches only, nests and
    restores; the process-wide mode is what a thread without an override sees (no GPU needed: the mode is host state)."""
    import threading
    from mdqe_cvpr2023_amd import _lib, ops
    h = _lib.load_library()
    assert h.mdqe_set_gemm_precision_thread(3) != 0 and h.mdqe_set_gemm_precision_thread(-2) != 0       # EINVAL, nothing changed
    assert ops.get_gemm_precision() == "f32"
    seen = {}

    def other():
        seen["other"] = ops.get_gemm_precision()

    with ops.gemm_precision("f16x3"):
        assert ops.get_gemm_precision() == "f16x3"
        t = threading.Thread(target=other); t.start(); t.join()
        with ops.gemm_precision("f32"):
            assert ops.get_gemm_precision() == "f32"
        assert ops.get_gemm_precision() == "f16x3"
        with pytest.raises(RuntimeError):
            with ops.gemm_precision("f32"):
                raise RuntimeError("boom")
        assert ops.get_gemm_precision() == "f16x3"                # restored on the way out of a failing block too
    assert ops.get_gemm_precision() == "f32" and seen["other"] == "f32"
    ops.set_gemm_precision("f16x3")
    try:
        with ops.gemm_precision("f32"):
            assert ops.get_gemm_precision() == "f32"
            t = threading.Thread(target=other); t.start(); t.join()
        assert seen["other"] == "f16x3" and ops.get_gemm_precision() == "f16x3"
    finally:
        ops.set_gemm_precision("f32")
