"""bench.py's choice of counter evidence, and the test utility behind the sharded-run comparison (no GPU).

`roofline.traffic` is quoted only from counter passes taken on the engine sources the run uses (a digest travels with
profiles/*/pmc_summary.json).  A round leaves several profile directories; the passes of the final build need not be the
last by name (round 4: r04_final sorts before r04_units)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _summary(digest, fetch, write):
    k = "sg::emit_fast_kernel<true, false>(sg::DevProfile, sg::DevBatch)"
    return {"_meta": {"engine_source_digest": digest, "emit_fast_kernel_avg_ms": 2.4},
            k: {"FETCH_SIZE": {"mean_per_dispatch": fetch}, "WRITE_SIZE": {"mean_per_dispatch": write}, "SQ_INSTS_VALU": {"mean_per_dispatch": 1.2e9}}}


def test_counter_passes_of_this_build_are_preferred(tmp_path, monkeypatch):
    import bench
    from simuscop_amd.build import engine_source_digest
    now = engine_source_digest()
    for name, digest, fetch in (("r09_final", now, 100.0), ("r09_units", "0" * 16, 200.0), ("r09_aaa", "1" * 16, 300.0)):
        d = tmp_path / "profiles" / name
        d.mkdir(parents=True)
        (d / "pmc_summary.json").write_text(json.dumps(_summary(digest, fetch, 1000.0)))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    ev = bench.pmc_evidence()
    assert ev["source"] == os.path.join("profiles", "r09_final", "pmc_summary.json") and ev["digest"] == now
    assert ev["traffic"] == (2.0 * 100.0 + 1000.0) * 1024.0   # FETCH_SIZE doubled (gfx950), both in KiB
    # without passes of this build: the last directory by name, whose digest the caller then refuses as stale
    (tmp_path / "profiles" / "r09_final" / "pmc_summary.json").unlink()
    ev = bench.pmc_evidence()
    assert ev["source"] == os.path.join("profiles", "r09_units", "pmc_summary.json") and ev["digest"] != now
    # no passes at all
    for name in ("r09_units", "r09_aaa"):
        (tmp_path / "profiles" / name / "pmc_summary.json").unlink()
    assert bench.pmc_evidence() is None


def test_committed_counter_passes_name_their_sources():
    """Every committed summary says which engine sources it was taken on (the final directory of a round is regenerated
    whenever csrc/ changes: tools/final_round.sh)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.json")))
    assert paths
    with_digest = [p for p in paths if json.load(open(p)).get("_meta", {}).get("engine_source_digest")]
    assert any("r04_final" in p for p in with_digest)


def test_fastq_record_digest_is_order_independent(oracle_lib, tmp_path):
    lib = oracle_lib
    lib.orc_fastq_record_digest.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.orc_fastq_record_digest.restype = C.c_int
    recs = [b"@r%d/1\n%s\n+\n%s\n" % (i, b"ACGT"[i % 4:i % 4 + 1] * (20 + i % 7), b"J" * (20 + i % 7)) for i in range(1000)]

    def digest(*files):
        arr = (C.c_char_p * len(files))(*[str(f).encode() for f in files])
        n, s = C.c_uint64(), C.c_uint64()
        rc = lib.orc_fastq_record_digest(arr, len(files), C.byref(n), C.byref(s))
        return rc, n.value, s.value
    (tmp_path / "whole.fq").write_bytes(b"".join(recs))
    (tmp_path / "a.fq").write_bytes(b"".join(recs[600:]))
    (tmp_path / "b.fq").write_bytes(b"".join(reversed(recs[:600])))
    (tmp_path / "empty.fq").write_bytes(b"")
    whole = digest(tmp_path / "whole.fq")
    assert whole[0] == 0 and whole[1] == 1000
    assert digest(tmp_path / "a.fq", tmp_path / "empty.fq", tmp_path / "b.fq") == whole
    changed = list(recs)
    changed[17] = changed[17].replace(b"J", b"F", 1)
    (tmp_path / "changed.fq").write_bytes(b"".join(changed))
    assert digest(tmp_path / "changed.fq")[2] != whole[2]
    (tmp_path / "cut.fq").write_bytes(b"".join(recs)[:-5])
    assert digest(tmp_path / "cut.fq")[0] == -2
    assert digest(tmp_path / "missing.fq")[0] == -1
