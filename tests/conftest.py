import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Build (if needed) and load the CPU oracle -- test infrastructure only."""
    import ctypes
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "liboracle.so")
    srcs = [os.path.join(odir, f) for f in ("oracle.cpp", "oracle.h", "philox.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", odir, "liboracle.so", "oracle_cli"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    c = ctypes
    lib.orc_simulate.argtypes = [c.c_char_p, c.c_int, c.c_uint64, c.c_uint64, c.c_char_p, c.c_int]
    lib.orc_simulate.restype = c.c_int
    lib.orc_last_error.restype = c.c_char_p
    lib.orc_last_read_count.restype = c.c_uint64
    lib.orc_profile_load.argtypes = [c.c_char_p, c.c_int, c.c_int]
    lib.orc_profile_load.restype = c.c_void_p
    lib.orc_profile_free.argtypes = [c.c_void_p]
    lib.orc_profile_info.argtypes = [c.c_void_p, c.c_int]
    lib.orc_profile_info.restype = c.c_int
    lib.orc_profile_rate.argtypes = [c.c_void_p, c.c_int]
    lib.orc_profile_rate.restype = c.c_double
    lib.orc_profile_array.argtypes = [c.c_void_p, c.c_int]
    lib.orc_profile_array.restype = c.POINTER(c.c_double)
    lib.orc_profile_kmer.argtypes = [c.c_void_p, c.c_int, c.c_char_p]
    lib.orc_profile_indel_gaps.argtypes = [c.c_void_p, c.POINTER(c.c_uint64), c.POINTER(c.c_uint64), c.c_int]
    lib.orc_profile_indel_gaps.restype = c.c_int
    lib.orc_predict_philox.argtypes = [c.c_void_p, c.c_char_p, c.c_int, c.c_int, c.c_uint64, c.c_uint32,
                                       c.c_uint32, c.c_char_p, c.c_char_p]
    lib.orc_predict_philox.restype = c.c_int
    lib.orc_philox4x32_10.argtypes = [c.POINTER(c.c_uint32), c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)]
    return lib
