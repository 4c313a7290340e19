import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle, build
    build(ref=False)
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference itself (oracle/_ref), when it has been built in this container."""
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def harness():
    """Device functions of seeq_kernel_core.h compiled for the host."""
    so = os.path.join(ROOT, "tests", "libharness.so")
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    core = os.path.join(ROOT, "seeq_amd", "csrc", "seeq_kernel_core.h")
    dfa = os.path.join(ROOT, "seeq_amd", "csrc", "seeq_dfa.h")
    plan = os.path.join(ROOT, "seeq_amd", "csrc", "seeq_plan.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core), os.path.getmtime(dfa), os.path.getmtime(plan)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", src, "-o", so])
    H = C.CDLL(so)
    H.harness_scan.restype = C.c_long
    H.harness_scan.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_size_t]
    H.harness_compile.restype = C.c_int
    H.harness_compile.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    return H


@pytest.fixture(scope="session")
def string_cases():
    with open(os.path.join(GOLDEN, "ref_string_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cli_cases():
    with open(os.path.join(GOLDEN, "ref_cli_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def capi():
    """The product library (built in-tree)."""
    from seeq_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        _capi.build()
    _capi.lib()
    return _capi


@pytest.fixture(scope="session")
def gpu(capi):
    """GPU tests must run the HIP path: no device -> fail, never skip to a fallback."""
    n = capi.lib().seeqdevDeviceCount()
    assert n >= 1, "no HIP device visible: -m gpu tests need an MI355X"
    return n
