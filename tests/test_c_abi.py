"""The C-ABI library loads and exports every symbol include/amfwd.h declares (no compute calls: no GPU needed)."""
import os
import re
import subprocess

import pytest

from nnacousticmodeling_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "amfwd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = _declared_functions()
    assert len(names) >= 30 and "am_forward_ff" in names and "am_forward_lstm" in names
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), "libamfwd.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (am_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_no_torch_or_framework_types_in_the_abi():
    text = open(HEADER).read()
    assert "torch" not in text.lower() and "#include <hip" not in text
    assert 'extern "C"' in text


def test_library_metadata_and_error_path_without_gpu():
    lib = _lib.load()
    assert lib.am_version().startswith(b"amfwd")
    assert lib.am_kernel_name(0) == b"splice_ft" and lib.am_kernel_name(99) is None
    if lib.am_device_count() == 0:
        from nnacousticmodeling_amd.engine import Engine
        with pytest.raises(_lib.AmError) as ei:
            Engine(0)  # fails loudly: there is no CPU fallback
        assert ei.value.code == -19 and "no CPU fallback" in str(ei.value)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libamfwd.so"))
    with pytest.raises(ImportError) as ei:
        _lib.load()
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nnacousticmodeling_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("self-oracle", ""), "%s mentions the oracle" % f


def test_product_package_imports_no_torch():
    """north_star: host code is Python over a C ABI, "no PyTorch".  torch.distributed serves bench.py's barrier only and lives
    beside it (bench_ranks.py), outside the package."""
    import re
    pkg = os.path.join(ROOT, "nnacousticmodeling_amd")
    for name in sorted(os.listdir(pkg)):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert not re.search(r"^\s*(import|from)\s+(torch|chainer|triton)\b", src, re.M), name


def test_stream_order_of_the_recurrent_weights():
    """The second copy of the cells' / wide LSTMs' packed weights (round 4): element by element the fragment map the kernels assume --
    out[block][batch][wave][sub-step][lane][j] = w[64 block + 16 (wave & 3) + (lane & 15)][128 batch + 64 (wave >> 2) + 16 sub-step + 4 (lane >> 4) + j]
    -- a permutation (nothing lost, nothing twice), and the shapes it refuses.  Host code only: no GPU."""
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for rows, ld in ((64, 128), (256, 384), (128, 1024)):
        w = rng.standard_normal((rows, ld)).astype(np.float32)
        out = np.zeros(rows * ld, np.float32)
        assert lib.am_pack_stream_order(w.ctypes.data_as(C.POINTER(C.c_float)), rows, ld, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
        o = out.reshape(rows // 64, ld // 128, 8, 4, 64, 4)
        blk, bat, wave, sub, lane, j = np.meshgrid(*[np.arange(n) for n in o.shape], indexing="ij")
        want = w[64 * blk + 16 * (wave & 3) + (lane & 15), 128 * bat + 64 * (wave >> 2) + 16 * sub + 4 * (lane >> 4) + j]
        assert np.array_equal(o, want)
        assert np.array_equal(np.sort(out), np.sort(w.ravel()))
    bad = np.zeros((60, 128), np.float32)
    assert lib.am_pack_stream_order(bad.ctypes.data_as(C.POINTER(C.c_float)), 60, 128, bad.ctypes.data_as(C.POINTER(C.c_float))) != 0
    assert lib.am_pack_stream_order(bad.ctypes.data_as(C.POINTER(C.c_float)), 64, 96, bad.ctypes.data_as(C.POINTER(C.c_float))) != 0
