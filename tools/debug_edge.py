"""Run one tests/edge_inputs.py configuration through the oracle (Philox mode) and the GPU path; print where the files part.
usage (GPU box): python tools/debug_edge.py fasta_repeated_name"""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import edge_inputs  # noqa: E402
import gpu_run  # noqa: E402

SEED = 0x5EED0E
name = sys.argv[1]
wd = tempfile.mkdtemp(prefix="dbg_")
cfg = edge_inputs.build(name, wd)
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
orc.orc_simulate.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int]
odir, gdir = wd + "/o", wd + "/g"
print("oracle rc", orc.orc_simulate(cfg.encode(), 1, SEED >> 32, SEED & 0xFFFFFFFF, odir.encode(), 4))
print("gpu", gpu_run.run_gpu(cfg, SEED, gdir))
for f in sorted(os.listdir(odir)):
    a, b = open(os.path.join(odir, f), "rb").read(), open(os.path.join(gdir, f), "rb").read()
    print(f, len(a), len(b), a == b)
    if a != b:
        i = next(k for k in range(min(len(a), len(b))) if a[k] != b[k])
        s = a.rfind(b"\n@", 0, i) + 1
        print("first difference at", i, "record starts at", s)
        print("oracle:", a[s:s + 700].decode(errors="replace"))
        print("gpu   :", b[s:s + 700].decode(errors="replace"))
        la, lb = a.split(b"\n"), b.split(b"\n")
        print("records", len(la) // 4, len(lb) // 4)
        names_a = [x for x in la[0::4]][:5], [x for x in lb[0::4]][:5]
        print(names_a)
        nd = [k for k in range(0, min(len(la), len(lb)), 4) if la[k] != lb[k]]
        print("differing names:", len(nd), "first at record", nd[0] // 4 if nd else None)
