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
