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
    config.addinivalue_line("markers", "long_oracle: oracle(mt) cases of 20-200 s, run in the background from session start")


_BG = {}


def _start_background_oracle(names):
    """tests/bg_oracle.py on the named cases, once per session (test infrastructure)."""
    if _BG or not names:
        return
    import tempfile
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "liboracle.so", "oracle_cli"], stdout=subprocess.DEVNULL)
    d = tempfile.mkdtemp(prefix="long_oracle_")
    workers = max(1, min(len(names), (os.cpu_count() or 2) - 1))
    _BG["dir"] = d
    _BG["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bg_oracle.py"), d, str(workers), *names],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)


def _long_names(items):
    return [it.callspec.params["name"] for it in items if it.get_closest_marker("long_oracle") and hasattr(it, "callspec")]


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: 1 if it.get_closest_marker("long_oracle") else 0)   # stable: the long cases are collected last


def pytest_collection_finish(session):
    """Start the long oracle(mt) runs as soon as the selection is known; the other tests run meanwhile."""
    if not session.config.option.collectonly:
        _start_background_oracle(_long_names(session.items))


def pytest_sessionfinish(session, exitstatus):
    import shutil
    p = _BG.get("proc")
    if p is not None:
        if p.poll() is None:
            import signal
            os.killpg(p.pid, signal.SIGTERM)   # the process group started above (runner + its oracle children), nothing else
        shutil.rmtree(_BG["dir"], ignore_errors=True)


@pytest.fixture(scope="session")
def long_oracle_runs(request):
    _start_background_oracle(_long_names(request.session.items))
    return _BG


@pytest.fixture(scope="session")
def oracle_lib():
    """Build (if needed) and load the CPU oracle -- test infrastructure only."""
    import ctypes
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "liboracle.so")
    srcs = [os.path.join(odir, f) for f in ("oracle.cpp", "train_oracle.cpp", "oracle.h", "philox.h")]
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
