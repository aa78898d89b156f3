"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/sdplr_hip.h declares,
the ctypes signature table matches the header, and — with no GPU — refuses to compute."""
import ctypes as C
import os
import re

import pytest

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(path, prefix):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int32_t|const char\*|void)\s+" + prefix + r"(\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        args = [] if args in ([""], ["void"]) else args
        out[m.group(1)] = len(args)
    return out


def test_header_matches_signature_table():
    decl = header_functions(os.path.join(ROOT, "include", "sdplr_hip.h"), "sdplr_hip_")
    assert set(decl) == set(cabi.SIGNATURES), set(decl) ^ set(cabi.SIGNATURES)
    for name, nargs in decl.items():
        assert nargs == len(cabi.SIGNATURES[name]), name


def test_oracle_header_mirrors_hip_header():
    a = header_functions(os.path.join(ROOT, "include", "sdplr_hip.h"), "sdplr_hip_")
    b = header_functions(os.path.join(ROOT, "oracle", "sdplr_oracle.h"), "sdplr_oracle_")
    for name, nargs in a.items():
        assert b.get(name) == nargs, name


def test_library_loads_and_exports_every_symbol(hip_abi):
    for name in cabi.SIGNATURES:
        assert hasattr(hip_abi.lib, "sdplr_hip_" + name)
    assert "gfx950" in hip_abi.version_string()


def test_no_cpu_fallback(hip_abi):
    n = C.c_int32(-1)
    assert hip_abi.device_count(C.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sj.SdplrError) as e:
        sj.DeviceSolver(hip_abi, 4, 4, 2, 4)
    assert e.value.code == cabi.ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "sdplrplus.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for bad in ("import oracle", "from oracle", "sdplr_oracle", "oracle/", "oracle.py"):
                    assert bad not in src, (f, bad)
