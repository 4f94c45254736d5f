"""CPU-side checks of the boundary: the library loads, exports every symbol of include/pp.h,
and the pure-host entry points agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np

import oracle_lib as O
from path_planner_amd import _abi as A
from path_planner_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = A.load_library()
    hdr = open(os.path.join(ROOT, "include", "pp.h")).read()
    declared = set(re.findall(r"\b(pp_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pp_ctx", "pp_map"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), name
    # and the ctypes table covers the header
    assert declared == set(lib._pp_symbols)


def test_rust_sys_covers_the_header():
    """rust/path_planner_hip/src/sys.rs (never compiled here: the image has no rustc) declares exactly
    the functions of include/pp.h, each with as many arguments as the header gives it."""
    hdr = open(os.path.join(ROOT, "include", "pp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(ROOT, "rust", "path_planner_hip", "src", "sys.rs")).read()
    rs = re.sub(r"//[^\n]*", "", rs)

    def arity(args):
        args = args.strip()
        return 0 if args in ("", "void") else args.count(",") + 1

    c_fns = {m.group(1): arity(m.group(2)) for m in re.finditer(r"\b(pp_[a-z_0-9]+)\s*\(([^()]*)\)\s*;", hdr)}
    rs_fns = {m.group(1): arity(m.group(2).rstrip().rstrip(","))
              for m in re.finditer(r"pub fn (pp_[a-z_0-9]+)\(([^()]*)\)", rs)}
    assert len(c_fns) >= 50
    assert set(c_fns) == set(rs_fns), set(c_fns) ^ set(rs_fns)
    for name, n in c_fns.items():
        assert rs_fns[name] == n, (name, n, rs_fns[name])


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(A.Car) == 32 and C.sizeof(A.Config) == 128
    assert C.sizeof(A.Param) == 72 and C.sizeof(A.RsWord) == 56
    assert C.sizeof(A.Path) == 56 and C.sizeof(A.ExpandOut) == 64


def test_strerror_and_version():
    lib = A.load_library()
    assert b"gfx950" in lib.pp_version()
    assert lib.pp_strerror(A.PP_NO_PATH).startswith(b"no path")
    assert lib.pp_strerror(A.PP_ERR_DEGENERATE_RS)


def test_motion_set_matches_oracle():
    conf = A.demo_config()
    s0, d0 = O.motion_set(conf)
    s1, d1 = api.calc_motion_set(conf)
    assert (s0 == s1).all() and (d0 == d1).all() and len(s1) == 78
    conf.n_steer = 3
    s0, d0 = O.motion_set(conf)
    s1, d1 = api.calc_motion_set(conf)
    assert (s0 == s1).all() and (d0 == d1).all() and len(s1) == 10


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        A.load_library(str(tmp_path / "libnope.so"))


def test_cpp_mirror_header_compiles():
    """include/path_planner.hpp (the reference's names over the C ABI) is valid C++17 on its own."""
    import subprocess
    src = "#include \"path_planner.hpp\"\nint main() { return 0; }\n"
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++",
                        "-I", os.path.join(ROOT, "include"), "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr
