"""CPU: the C-ABI library loads without a GPU and exports every symbol include/lto.h declares; the Python
binding table covers the same set; creating a context without a device fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "lto.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lto_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(lto.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "liblto_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms     # the ctypes table binds exactly the header's entry points


def test_version_and_error_codes():
    lib = lto.load_library()
    assert lib.lto_version() == 100
    assert (_lib.LTO_EINVAL, _lib.LTO_ENULL, _lib.LTO_EUNSUPPORTED) == (-1, -2, -3)
    assert lib.lto_create(None, 0) == _lib.LTO_ENULL


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lto.LtoError) as ei:
        lto.Context(0)
    assert ei.value.code == _lib.LTO_ENODEVICE


def test_product_never_touches_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import/link/execute oracle/."""
    pkg = os.path.join(ROOT, "lowthrustopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "lto_oracle" not in text and "liblto_oracle" not in text, f
                assert not re.search(r"#include\s+[\"<].*oracle", text), f


def _julia_ccalls():
    """(symbol, number of argument types) of every ccall in julia/LowThrustOptHIP.jl."""
    text = open(os.path.join(ROOT, "julia", "LowThrustOptHIP.jl")).read()
    out = []
    for m in re.finditer(r"ccall\(\s*(\(:(lto_[a-z0-9_]+), liblto\)|entry\(ctx, :([a-z_]+)\))\s*,\s*[A-Za-z{}]+\s*,\s*\(", text):
        name = m.group(2) or ("lto_" + m.group(3))
        depth, i = 1, m.end()
        while depth:                      # the argument-type tuple, up to its closing parenthesis
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        types = text[m.end():i - 1]
        n = 0 if not types.strip().strip(",") else len([t for t in re.split(r",(?![^{]*\})", types) if t.strip()])
        out.append((name, n))
    return out


def test_every_julia_ccall_has_a_ctypes_twin_with_the_same_arity():
    """The Julia glue cannot run here (no julia binary); what can be checked statically is that every `ccall` names an
    exported entry point and passes as many arguments as the ctypes prototype the GPU tests execute."""
    calls = _julia_ccalls()
    assert len(calls) >= 35
    lib = ctypes.CDLL(lto.LIB_PATH)
    for name, nargs in calls:
        assert name in _lib.SIGNATURES, "julia ccall of %s has no ctypes twin" % name
        assert hasattr(lib, name)
        assert nargs == len(_lib.SIGNATURES[name][1]), "%s: julia passes %d arguments, the ctypes twin %d" % (name, nargs, len(_lib.SIGNATURES[name][1]))
    # the group forms share the argument lists of the single-context sweeps
    for sweep in ("indirect_defect", "indirect_jacobian", "direct_defect", "direct_jacobian"):
        assert len(_lib.SIGNATURES["lto_" + sweep][1]) == len(_lib.SIGNATURES["lto_group_" + sweep][1])
