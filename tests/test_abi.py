"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/rmpc.h declares; product code fails loudly without a device.  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from robot_mpcs_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "rmpc.h")).read()
    declared = set(re.findall(r"\b(rmpc_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = C.CDLL(lib.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/rmpc.h but not exported"
    assert declared == set(lib.EXPORTED_SYMBOLS)


def test_descriptor_layout_and_version(lib):
    L = lib.load_library()
    assert L.rmpc_desc_size() == C.sizeof(lib.RmpcDesc)
    # the macro of include/rmpc.h is what the library reports (0.2.0: rmpc_retarget struct, rmpc_advance_obstacles_device)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rmpc.h")).read()
    ver = int(re.search(r"#define RMPC_VERSION (\d+)", hdr).group(1))
    assert L.rmpc_version() == ver == 201
    assert [L.rmpc_kernel_name(i).decode() for i in range(5)] == ["k_pack", "k_sweep", "k_riccati", "k_step", "k_unpack"]


def test_workspace_bytes_is_host_only_and_scales(lib):
    from robot_mpcs_amd.scenarios import make_scenario
    L = lib.load_library()
    sc = make_scenario("cfg2", B=1)
    d = lib.make_desc(sc.desc)
    w1 = L.rmpc_workspace_bytes(C.byref(d), 64)
    w2 = L.rmpc_workspace_bytes(C.byref(d), 4096)
    assert 0 < w1 < w2
    assert w2 < 2 * 64 * w1
    # cfg2 at B = 4096: a few hundred MB, far below 288 GB of HBM
    assert 100e6 < w2 < 1e9


def test_invalid_descriptors_are_rejected(lib):
    from robot_mpcs_amd.scenarios import make_scenario
    L = lib.load_library()
    sc = make_scenario("cfg1", B=1)
    d = lib.make_desc(sc.desc)
    d.struct_size = 4
    assert L.rmpc_workspace_bytes(C.byref(d), 8) == -1
    d = lib.make_desc(sc.desc)
    d.off_goal = 10_000
    assert L.rmpc_workspace_bytes(C.byref(d), 8) == -1
    assert b"goal" in L.rmpc_last_error()
    d = lib.make_desc(sc.desc)
    d.module_kind[0] = 77
    assert L.rmpc_workspace_bytes(C.byref(d), 8) == -1


def test_no_silent_cpu_fallback(lib):
    """Without a HIP device the product raises; it never routes to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from robot_mpcs_amd.scenarios import make_scenario
    sc = make_scenario("cfg1", B=1)
    with pytest.raises(lib.RmpcError):
        lib.Solver(sc.desc, max_batch=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "robot_mpcs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", src), f
                assert "librmpc_oracle" not in src and "rmpc_oracle.h" not in src, f
