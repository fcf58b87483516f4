"""The C-ABI shared library loads without a GPU, exports every symbol include/bls381_hip.h declares, and
refuses to work without a device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest
from blsutil import ROOT

import bls381_amd as B

HEADER = os.path.join(ROOT, "include", "bls381_hip.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bls381_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.exists(B.LIB_PATH), "run `python __graft_entry__.py` (build()) first"
    lib = ctypes.CDLL(B.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(B.SYMBOLS) == declared, "python binding list out of sync with the header"


def test_error_strings():
    lib = B.load_library()
    assert lib.bls381_strerror(0) == b"ok"
    assert b"no CPU path" in lib.bls381_strerror(2)


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(B.Bls381Error, match="no usable HIP device"):
        B.Engine(0)


def test_product_sources_never_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import, link or open it."""
    pkg = os.path.join(ROOT, "plonky2-bls12-381-pairing_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hip.h", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyref" not in txt and "bls381_oracle" not in txt and "hostsim" not in txt.replace("tests/hostsim", ""), os.path.join(dirpath, f)


def test_cpp_host_mirror_compiles():
    """plonky2-bls12-381-pairing_amd/host/bls381_pairing.hpp (C++ mirror of the reference interface) is well-formed
    and links against the library."""
    import subprocess, tempfile
    src = os.path.join(ROOT, "tests", "host_mirror_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "host_mirror_check")
        subprocess.check_call(["g++", "-std=c++17", src, "-o", exe, "-L", os.path.dirname(B.LIB_PATH), "-lbls381_hip",
                               "-Wl,-rpath," + os.path.dirname(B.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
        assert subprocess.call([exe]) == 0      # without a GPU the Backend constructor throws PairingError -> exit 0


def test_fq12_order_helpers_host_only():
    """bls381_fq12_ark_to_myfq12 / _myfq12_to_ark are pure host permutations (no GPU): check them against the oracle's
    statement of the MyFq12 order (src/fields/helpers.rs:22-76) and as inverses of each other."""
    import random
    import numpy as np
    from blsutil import O, enc_fp12s, enc_fps, rand_fp12
    lib = B.load_library()
    rng = random.Random(12)
    xs = [rand_fp12(rng) for _ in range(5)]
    ark = enc_fp12s(xs)
    my = np.zeros_like(ark)
    back = np.zeros_like(ark)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    lib.bls381_fq12_ark_to_myfq12(ark.ctypes.data_as(u64p), my.ctypes.data_as(u64p), len(xs))
    assert np.array_equal(my, enc_fps([c for x in xs for c in O.fp12_to_myfq12(x)]))
    lib.bls381_fq12_myfq12_to_ark(my.ctypes.data_as(u64p), back.ctypes.data_as(u64p), len(xs))
    assert np.array_equal(back, ark)


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md shows the reference-side (Rust `extern "C"`) binding of EVERY function include/bls381_hip.h declares."""
    import re
    hdr = open(os.path.join(ROOT, "include", "bls381_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"^(?:int|void|float|const char\*|bls381_ctx\*)\s+(bls381_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 50
    bound = set(re.findall(r"fn (bls381_[a-z0-9_]+)\s*\(", doc))
    assert declared - bound == set(), sorted(declared - bound)
