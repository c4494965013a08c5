"""Build the native pieces in-tree (no JIT cache: the .so files travel with the repo snapshot).

  libsimuscop_amd.so  -- C ABI engine: HIP kernels for gfx950 + table conversion   (hipcc)
  libsimuscop_host.so -- C++ host side mirroring SimuSCoP's config/CLI surface      (g++)
  simuReads           -- `simuReads <config.txt>` command line, links the two above  (g++)
  seqToProfile        -- the profile trainer's command line (host/train_main.cpp)     (g++)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
INCLUDE = os.path.join(ROOT, "include")

ENGINE_SRCS = ["sg_kernels.hip", "sg_haplotypes.hip", "sg_deflate.hip", "sg_train.hip", "sg_api.cpp", "sg_tables.cpp", "sg_deflate.cpp"]
HOST_SRCS = ["host/config.cpp", "host/profile.cpp", "host/fasta.cpp", "host/variants.cpp", "host/genome.cpp",
             "host/simulate.cpp", "host/train.cpp"]
CLI_SRCS = ["host/main.cpp"]
TRAIN_CLI_SRCS = ["host/train_main.cpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_headers():
    out = [os.path.join(INCLUDE, "simuscop_amd.h")]
    for d, _, fs in os.walk(CSRC):
        out += [os.path.join(d, f) for f in fs if f.endswith(".h")]
    return out


def hipcc_path():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found: the engine is HIP-only (no CPU build)")


def build_engine(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    so = os.path.join(LIBDIR, "libsimuscop_amd.so")
    srcs = [os.path.join(CSRC, s) for s in ENGINE_SRCS]
    if force or _newer(so, srcs + _all_headers()):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-x", "hip"] + srcs + ["-o", so]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so


def build_host(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    so = os.path.join(LIBDIR, "libsimuscop_host.so")
    exe = os.path.join(LIBDIR, "simuReads")
    srcs = [os.path.join(CSRC, s) for s in HOST_SRCS]
    hdrs = _all_headers()
    if force or _newer(so, srcs + hdrs):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-pthread", "-I", INCLUDE,
               "-I", CSRC] + srcs + ["-o", so, "-L", LIBDIR, "-lsimuscop_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli = [os.path.join(CSRC, s) for s in CLI_SRCS]
    if force or _newer(exe, cli + hdrs + [so]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I", INCLUDE, "-I", CSRC] + cli + \
              ["-o", exe, "-L", LIBDIR, "-lsimuscop_host", "-lsimuscop_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    texe = os.path.join(LIBDIR, "seqToProfile")
    tcli = [os.path.join(CSRC, s) for s in TRAIN_CLI_SRCS]
    if force or _newer(texe, tcli + hdrs + [so]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I", INCLUDE, "-I", CSRC] + tcli + \
              ["-o", texe, "-L", LIBDIR, "-lsimuscop_host", "-lsimuscop_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so, exe


def engine_source_digest():
    """sha256 over the engine's sources (csrc/*.hip, *.cpp, *.h; the C ABI header): what counters and traces were taken
    on.  tools/profile_round.sh stores it with the counters, bench.py quotes counters only of the sources it runs."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(INCLUDE, "simuscop_amd.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_all(force=False, verbose=False):
    eng = build_engine(force, verbose)
    host, exe = build_host(force, verbose)
    return eng, host, exe


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
