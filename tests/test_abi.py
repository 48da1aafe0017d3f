"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports exactly what include/fakequant.h
declares, and the product path refuses to run without a HIP device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "fakequant.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fq_[a-z0-9_]+)\s*\(", text)))


def test_library_built_and_exports_every_declared_symbol():
    from quantization.mxnet_amd import _lib
    path = _lib.library_path()
    assert os.path.exists(path), "libfakequant.so missing: run python -m quantization.mxnet_amd.csrc.build"
    lib = ctypes.CDLL(path)
    declared = _declared()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert sorted(_lib.EXPORTS) == declared, "ctypes table and header disagree"


def test_version_and_error_string_without_gpu():
    from quantization.mxnet_amd import _lib
    assert _lib.LIB.fq_version() >= 100
    assert isinstance(_lib.LIB.fq_last_error(), bytes)
    assert _lib.LIB.fq_act_workspace_bytes(128) >= 128 * 4
    assert _lib.LIB.fq_kl_workspace_bytes(53, 2048) >= 53 * 2048 * 8


def test_argument_validation_happens_before_any_launch():
    from quantization.mxnet_amd import _lib
    rc = _lib.LIB.fq_fake_quant_online(None, None, 1, 1, 8, 0, None, None, None, None)
    assert rc == 1
    assert b"null pointer" in _lib.LIB.fq_last_error()
    rc = _lib.LIB.fq_kl_search(ctypes.c_void_p(8), 1, 2048, 256, 128, ctypes.c_void_p(8), ctypes.c_void_p(8), None)
    assert rc == 1 and b"min_bins should be greater than levels" in _lib.LIB.fq_last_error()


def test_product_refuses_cpu_tensors():
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd._lib import FakeQuantError
    x = torch.zeros(2, 3, 4, 4)
    with pytest.raises(FakeQuantError, match="no CPU fallback"):
        ops.fake_quant_online(x)
    with pytest.raises(FakeQuantError):
        ops.weight_fake_quant(x, 2)


def test_missing_library_fails_loudly(tmp_path):
    from quantization.mxnet_amd import _lib
    missing = _lib.load(str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FakeQuantError, match="not available"):
        missing.fq_version()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "quantization")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_library_carries_the_id_of_the_sources_it_sits_beside(tmp_path):
    """csrc/build.py decides "up to date" by content: the sha1 of sources + headers + flags is compiled into the library
    (fq_build_id) and found again in the file's bytes; a changed source means a different id, whatever the file times say."""
    from quantization.mxnet_amd import _lib
    from quantization.mxnet_amd.csrc import build as B
    want = B.source_id()
    assert len(want) == 40
    if os.environ.get("FQ_LIB_PATH"):
        pytest.skip("a variant library is loaded")
    assert B.built_id(B.OUT) == want, "libfakequant.so was built from other sources than this tree: run __graft_entry__.build()"
    assert _lib.LIB.fq_build_id().decode() == want
    assert B.up_to_date()
    assert B.source_id(["-DFQ_SOMETHING=1"]) != want
    stale = tmp_path / "lib.so"
    stale.write_bytes(b"\x7fELF....FQ_BUILD_ID=" + b"0" * 40 + b"....")
    assert B.built_id(str(stale)) == "0" * 40 and not B.up_to_date(str(stale))
    assert B.built_id(str(tmp_path / "absent.so")) is None
