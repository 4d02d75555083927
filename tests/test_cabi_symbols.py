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
    assert lib.lto_version() == 102
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
    """(symbol, return type, [argument types]) of every ccall in julia/LowThrustOptHIP.jl."""
    text = open(os.path.join(ROOT, "julia", "LowThrustOptHIP.jl")).read()
    out = []
    for m in re.finditer(r"ccall\(\s*(\(:(lto_[a-z0-9_]+), liblto\)|entry\(ctx, :([a-z_]+)\))\s*,\s*([A-Za-z{}]+)\s*,\s*\(", text):
        name = m.group(2) or ("lto_" + m.group(3))
        depth, i = 1, m.end()
        while depth:                      # the argument-type tuple, up to its closing parenthesis
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        types = text[m.end():i - 1]
        args = [] if not types.strip().strip(",") else [t.strip() for t in re.split(r",(?![^{]*\})", types) if t.strip()]
        out.append((name, m.group(4), args))
    return out


def _julia_class(t):
    """Width class of a Julia ccall type."""
    if t in ("Cint",):
        return "i32"
    if t in ("Clong",):
        return "i64"
    if t in ("Csize_t",):
        return "size"
    if t in ("Cdouble",):
        return "f64"
    if t in ("Cvoid", "Nothing"):
        return "void"
    if t in ("DevPtr", "Cstring") or t.startswith(("Ptr{", "Ref{")):
        return "ptr"
    raise AssertionError("unclassified Julia ccall type %r" % t)


def _ctypes_class(t):
    if t is None:
        return "void"
    if t is ctypes.c_int:
        return "i32"
    if t is ctypes.c_long:
        assert ctypes.sizeof(ctypes.c_long) == 8
        return "i64"
    if t is ctypes.c_size_t:
        return "size"
    if t is ctypes.c_double:
        return "f64"
    if t in (ctypes.c_void_p, ctypes.c_char_p) or issubclass(t, ctypes._Pointer):
        return "ptr"
    raise AssertionError("unclassified ctypes type %r" % (t,))


def test_every_julia_ccall_has_a_ctypes_twin_with_the_same_types():
    """The Julia glue cannot run here (no julia binary); what can be checked statically is that every `ccall` names an
    exported entry point and passes, argument by argument, the same WIDTH CLASS (32-bit int, 64-bit long, size_t, double,
    pointer) as the ctypes prototype the GPU tests execute, and expects the same class back -- a `Cint` where the C function
    takes a `long` would pass an arity check and corrupt the call."""
    calls = _julia_ccalls()
    assert len(calls) >= 35
    lib = ctypes.CDLL(lto.LIB_PATH)
    for name, ret, args in calls:
        assert name in _lib.SIGNATURES, "julia ccall of %s has no ctypes twin" % name
        assert hasattr(lib, name)
        cret, cargs = _lib.SIGNATURES[name]
        assert len(args) == len(cargs), "%s: julia passes %d arguments, the ctypes twin %d" % (name, len(args), len(cargs))
        assert _julia_class(ret) == _ctypes_class(cret), "%s: return type %s vs %r" % (name, ret, cret)
        for k, (jt, ct) in enumerate(zip(args, cargs)):
            assert _julia_class(jt) == _ctypes_class(ct), "%s: argument %d is %s in Julia, %r in ctypes" % (name, k, jt, ct)
    # the group forms share the argument lists of the single-context sweeps
    for sweep in ("indirect_defect", "indirect_jacobian", "direct_defect", "direct_jacobian"):
        assert len(_lib.SIGNATURES["lto_" + sweep][1]) == len(_lib.SIGNATURES["lto_group_" + sweep][1])


def _header_prototypes():
    """name -> (return class, [argument classes]) parsed from include/lto.h."""
    text = open(os.path.join(ROOT, "include", "lto.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)

    def cls(decl):
        d = decl.strip()
        if "*" in d:
            return "ptr"
        base = re.sub(r"\b(const|unsigned)\b", "", d).split()
        base = base[0] if base else ""
        return {"int": "i32", "long": "i64", "size_t": "size", "double": "f64", "void": "void"}[base]
    out = {}
    for m in re.finditer(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(lto_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", text, flags=re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        out[name] = (cls(ret), [cls(re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", a) if not a.endswith("*") else a) for a in args])
    return out


def test_ctypes_table_matches_the_header_types():
    """The same width classes between include/lto.h and the ctypes table (the other half of the chain header -> ctypes -> Julia)."""
    protos = _header_prototypes()
    assert len(protos) >= 60
    for name, (cret, cargs) in _lib.SIGNATURES.items():
        assert name in protos, name
        hret, hargs = protos[name]
        assert hret == _ctypes_class(cret), "%s: returns %s in the header, %r in ctypes" % (name, hret, cret)
        assert hargs == [_ctypes_class(t) for t in cargs], "%s: header %s, ctypes %s" % (name, hargs, [_ctypes_class(t) for t in cargs])
