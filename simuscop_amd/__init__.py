"""simuscop_amd -- MI355X-native read-sampling engine behind SimuSCoP's simuReads surface.

Python is only glue here (ctypes over the C ABI in include/simuscop_amd.h and over the C++ host
library); the product is `lib/libsimuscop_amd.so` (HIP kernels, gfx950), `lib/libsimuscop_host.so`
and the `lib/simuReads` command line.  There is no Python or CPU compute path: loading fails loudly
when the native libraries are missing.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(PKG_DIR, "lib")
ENGINE_SO = os.path.join(LIB_DIR, "libsimuscop_amd.so")
HOST_SO = os.path.join(LIB_DIR, "libsimuscop_host.so")
SIMUREADS = os.path.join(LIB_DIR, "simuReads")

SG_K_NAMES = ["plan", "namebase", "indel", "scan", "emit", "emit_slow"]

# every symbol include/simuscop_amd.h declares
ENGINE_SYMBOLS = [
    "sg_create", "sg_destroy", "sg_last_error", "sg_set_stream", "sg_set_seed", "sg_set_strict_bases", "sg_load_profile",
    "sg_upload_haplotypes", "sg_reference_begin", "sg_reference_chunk", "sg_sync", "sg_reference_scan",
    "sg_reference_commit", "sg_build_haplotypes", "sg_haplotype_codes", "sg_compress", "sg_fetch_compressed",
    "sg_bgzf_eof", "sg_deflate_plan", "sg_detach_outputs", "sg_outputs_sizes", "sg_outputs_fetch",
    "sg_outputs_last_error", "sg_release_outputs", "sg_plan", "sg_sample", "sg_result", "sg_fetch", "sg_device_output",
    "sg_gc_percent", "sg_set_profiling", "sg_kernel_times", "sg_emit_info", "sg_emit_variant", "sg_cdf_count_le", "sg_fetch_range", "sg_host_alloc",
    "sg_sub_row_identity_first", "sg_row_symbols", "sg_alias_row", "sg_window_weights", "sg_windows_build", "sg_plan_windows", "sg_plan_range", "sg_windows_drop",
    "sg_host_free", "sg_profile_prepare", "sg_profile_tables_error", "sg_load_prepared_profile", "sg_profile_tables_free", "sg_train_count",
    "sg_train_begin", "sg_train_feed", "sg_train_capped", "sg_train_finish", "sg_train_end",
    "sg_release_cached_memory",
]


class SgTrainCounts(C.Structure):
    """sg_train_counts / orc_train_counts (same layout): caller-allocated count arrays + scalar counters."""
    _fields_ = [("subs1", C.POINTER(C.c_uint64)), ("subs2", C.POINTER(C.c_uint64)), ("kmers", C.POINTER(C.c_uint64)),
                ("quality", C.POINTER(C.c_uint64)), ("isize", C.POINTER(C.c_uint64)),
                ("ins_len", C.POINTER(C.c_uint64)), ("del_len", C.POINTER(C.c_uint64)),
                ("lines", C.c_uint64), ("reads_counted", C.c_uint64), ("cigar_chars", C.c_uint64), ("insert_events", C.c_uint64),
                ("delete_events", C.c_uint64), ("isize_overflow", C.c_uint64), ("indel_len_overflow", C.c_uint64),
                ("skipped_overhang", C.c_uint64), ("gc_rejected", C.c_uint64), ("gc_windows", C.c_uint64), ("capped", C.c_uint64)]


class SgTrainSetup(C.Structure):
    """sg_train_setup (include/simuscop_amd.h)"""
    _fields_ = [("contig_keys", C.POINTER(C.c_char_p)), ("n_contigs", C.c_uint32), ("bases", C.c_char_p), ("kmer", C.c_int32),
                ("bins", C.c_int32), ("n_isize", C.c_uint32), ("n_indel_len", C.c_uint32), ("count_gc", C.c_int32), ("window", C.c_uint32),
                ("max_reads", C.c_uint64),
                ("target_first", C.POINTER(C.c_uint64)), ("target_spos", C.POINTER(C.c_int64)), ("target_epos", C.POINTER(C.c_int64)),
                ("n_snv", C.c_uint64), ("snv_contig", C.POINTER(C.c_uint32)), ("snv_pos", C.POINTER(C.c_int64)), ("snv_alt", C.c_char_p),
                ("snv_homo", C.POINTER(C.c_uint8)),
                ("n_ins", C.c_uint64), ("ins_contig", C.POINTER(C.c_uint32)), ("ins_pos", C.POINTER(C.c_int64)), ("ins_len", C.POINTER(C.c_int32)),
                ("n_del", C.c_uint64), ("del_contig", C.POINTER(C.c_uint32)), ("del_pos", C.POINTER(C.c_int64)), ("del_len", C.POINTER(C.c_int32))]


class SgProfileCdf(C.Structure):
    _fields_ = [("n_bases", C.c_int32), ("bases", C.c_char * 8), ("kmer", C.c_int32), ("bins", C.c_int32),
                ("read_length", C.c_int32), ("n_qual", C.c_int32), ("min_qual", C.c_int32),
                ("insert_rate", C.c_double), ("del_rate", C.c_double),
                ("ins_cdf", C.POINTER(C.c_double)), ("n_ins", C.c_int32),
                ("del_cdf", C.POINTER(C.c_double)), ("n_del", C.c_int32),
                ("subs_cdf1", C.POINTER(C.c_double)), ("subs_cdf2", C.POINTER(C.c_double)),
                ("qual_cdf", C.POINTER(C.c_double)), ("isize_cdf", C.POINTER(C.c_double)),
                ("n_isize", C.c_int32), ("isize_min", C.c_int32), ("insert_size", C.c_int32)]


class SgWindow(C.Structure):
    _fields_ = [("hap_base", C.c_uint64), ("chain", C.c_uint32), ("spos", C.c_uint32), ("len", C.c_uint32),
                ("n_reads", C.c_int32), ("seg", C.c_uint32), ("slot_base", C.c_uint32)]


class SgBatch(C.Structure):
    _fields_ = [("batch_id", C.c_uint32), ("paired", C.c_int32), ("name_prefix", C.c_char_p),
                ("windows", C.POINTER(SgWindow)), ("n_windows", C.c_uint64),
                ("seg_size", C.POINTER(C.c_uint32)), ("seg_first_window", C.POINTER(C.c_uint32)),
                ("n_segs", C.c_uint32), ("first_window", C.c_uint32), ("first_slot", C.c_uint32)]


class SgGcWindow(C.Structure):
    _fields_ = [("start", C.c_uint64), ("chain", C.c_uint32), ("len", C.c_uint32)]


class SgContig(C.Structure):
    _fields_ = [("raw_offset", C.c_uint64), ("length", C.c_uint64), ("line_bases", C.c_uint32), ("line_width", C.c_uint32)]


class SgHapPiece(C.Structure):
    _fields_ = [("dst", C.c_uint64), ("src", C.c_uint64), ("len", C.c_uint32), ("chain", C.c_uint32),
                ("contig", C.c_uint32), ("kind", C.c_uint32)]


class SgHapPatch(C.Structure):
    _fields_ = [("dst", C.c_uint64), ("chain", C.c_uint32), ("base", C.c_uint32)]


class SimuOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("has_seed", C.c_int32), ("seed", C.c_uint64), ("write_files", C.c_int32),
                ("fetch", C.c_int32), ("quiet", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
                ("output_dir", C.c_char_p), ("repeat_sample", C.c_int32), ("host_haplotypes", C.c_int32), ("gzip", C.c_int32),
                ("shard_contigs", C.c_int32), ("no_eof_block", C.c_int32), ("exchange", C.c_void_p), ("exchange_user", C.c_void_p),
                ("crlf_as_lf", C.c_int32), ("strict_bases", C.c_int32), ("unique_contigs", C.c_int32)]


# simu_options.exchange: all-reduce(sum) of n doubles over the ranks, in place
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


class SimuStats(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("fragments", C.c_uint64), ("fastq_bytes", C.c_uint64),
                ("planned_reads", C.c_uint64), ("windows", C.c_uint64), ("segments", C.c_uint64),
                ("batches", C.c_uint64), ("t_load", C.c_double), ("t_haplotypes", C.c_double),
                ("t_plan", C.c_double), ("t_sample", C.c_double), ("t_fetch", C.c_double),
                ("t_write", C.c_double), ("t_total", C.c_double), ("kernel_ms", C.c_float * 8),
                ("queued_items", C.c_uint64), ("requeued_batches", C.c_uint64), ("t_engine", C.c_double),
                ("t_reference", C.c_double), ("t_hap_device", C.c_double), ("t_plan_api", C.c_double), ("t_compress", C.c_double), ("gz_bytes", C.c_uint64)]


_engine = None
_host = None


def load_engine():
    """dlopen libsimuscop_amd.so and declare the C ABI.  Raises if it has not been built."""
    global _engine
    if _engine is not None:
        return _engine
    if not os.path.exists(ENGINE_SO):
        raise RuntimeError(f"{ENGINE_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP engine is the only compute path)")
    lib = C.CDLL(ENGINE_SO, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.sg_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint64]
    lib.sg_destroy.argtypes = [vp]
    lib.sg_destroy.restype = None
    lib.sg_release_cached_memory.restype = None
    lib.sg_release_cached_memory.argtypes = []
    lib.sg_last_error.argtypes = [vp]
    lib.sg_last_error.restype = C.c_char_p
    lib.sg_set_stream.argtypes = [vp, vp]
    lib.sg_set_seed.argtypes = [vp, C.c_uint64]
    lib.sg_load_profile.argtypes = [vp, C.POINTER(SgProfileCdf)]
    lib.sg_upload_haplotypes.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    lib.sg_reference_begin.argtypes = [vp, C.c_uint64]
    lib.sg_reference_chunk.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_uint64]
    lib.sg_sync.argtypes = [vp]
    lib.sg_reference_scan.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.sg_reference_commit.argtypes = [vp, C.POINTER(SgContig), C.c_uint32]
    lib.sg_train_count.argtypes = [vp, C.c_char_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_int32, C.c_int32,
                                   C.c_uint32, C.c_uint32, C.POINTER(SgTrainCounts)]
    lib.sg_train_begin.argtypes = [vp, C.POINTER(SgTrainSetup)]
    lib.sg_train_feed.argtypes = [vp, C.c_char_p, C.c_uint64]
    lib.sg_train_capped.argtypes = [vp]
    lib.sg_train_finish.argtypes = [vp, C.POINTER(SgTrainCounts), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint64)]
    lib.sg_train_end.argtypes = [vp]
    lib.sg_train_end.restype = None
    lib.sg_build_haplotypes.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(SgHapPiece), C.c_uint64,
                                        C.c_char_p, C.c_uint64, C.POINTER(SgHapPatch), C.c_uint64]
    lib.sg_haplotype_codes.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p]
    lib.sg_compress.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.sg_fetch_compressed.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p]
    lib.sg_bgzf_eof.argtypes = [C.c_char_p]
    lib.sg_deflate_plan.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(C.c_uint32), C.c_char_p,
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
    lib.sg_deflate_plan.restype = C.c_uint32
    lib.sg_detach_outputs.argtypes = [vp, C.POINTER(vp)]
    lib.sg_outputs_sizes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.sg_outputs_fetch.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p]
    lib.sg_outputs_last_error.argtypes = [vp]
    lib.sg_outputs_last_error.restype = C.c_char_p
    lib.sg_release_outputs.argtypes = [vp, vp]
    lib.sg_plan.argtypes = [vp, C.POINTER(SgBatch)]
    lib.sg_sample.argtypes = [vp]
    lib.sg_result.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.sg_fetch.argtypes = [vp, vp, vp]
    lib.sg_fetch_range.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, vp]
    lib.sg_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    lib.sg_host_free.argtypes = [vp, vp]
    lib.sg_device_output.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.sg_gc_percent.argtypes = [vp, C.POINTER(SgGcWindow), C.c_uint64, C.POINTER(C.c_int32)]
    lib.sg_set_profiling.argtypes = [vp, C.c_int]
    lib.sg_kernel_times.argtypes = [vp, C.POINTER(C.c_float)]
    lib.sg_emit_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    lib.sg_emit_variant.argtypes = [vp]
    lib.sg_cdf_count_le.argtypes = [C.c_double]
    lib.sg_cdf_count_le.restype = C.c_uint64
    lib.sg_sub_row_identity_first.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
    lib.sg_row_symbols.argtypes = [C.POINTER(C.c_double), C.c_int]
    lib.sg_row_symbols.restype = C.c_uint32
    lib.sg_alias_row.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8),
                                 C.POINTER(C.c_uint8)]
    _engine = lib
    return lib


def load_host():
    """dlopen libsimuscop_host.so (C++ host side: config / profile / genome / driver)."""
    global _host
    if _host is not None:
        return _host
    load_engine()
    if not os.path.exists(HOST_SO):
        raise RuntimeError(f"{HOST_SO} is missing: run __graft_entry__.build()")
    lib = C.CDLL(HOST_SO, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.simu_default_options.argtypes = [C.POINTER(SimuOptions)]
    lib.simu_default_options.restype = None
    lib.simu_assign_contigs.argtypes = [C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.simu_assign_contigs.restype = None
    lib.simu_run.argtypes = [C.c_char_p, C.POINTER(SimuOptions), C.POINTER(SimuStats), C.c_char_p, C.c_size_t]
    lib.simu_selftest_haplotypes.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_size_t]
    lib.simu_open.argtypes = [C.c_char_p, C.POINTER(SimuOptions), C.POINTER(vp), C.c_char_p, C.c_size_t]
    lib.simu_close.argtypes = [vp]
    lib.simu_close.restype = None
    lib.simu_engine.argtypes = [vp]
    lib.simu_engine.restype = vp
    lib.simu_planned_reads.argtypes = [vp]
    lib.simu_planned_reads.restype = C.c_uint64
    lib.simu_chromosome_count.argtypes = [vp]
    lib.simu_weighted_length.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
    lib.simu_set_reads.argtypes = [vp, C.c_int, C.c_int64, C.c_char_p, C.c_size_t]
    lib.simu_prepare_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    lib.simu_get_stats.argtypes = [vp, C.POINTER(SimuStats)]
    lib.simu_get_stats.restype = None
    _host = lib
    return lib


class SimuError(RuntimeError):
    pass


def default_options(**kw) -> SimuOptions:
    o = SimuOptions()
    load_host().simu_default_options(C.byref(o))
    for k, v in kw.items():
        if k == "seed":
            o.has_seed, o.seed = 1, int(v)
        elif k == "output_dir":
            o.output_dir = v.encode() if isinstance(v, str) else v
        elif k == "exchange":
            o.exchange = C.cast(v, C.c_void_p).value   # an EXCHANGE_FN instance; the caller keeps it alive
        else:
            setattr(o, k, v)
    return o


def run_config(config_path: str, **opts) -> SimuStats:
    """`simuReads <config>` in-process (src/simuReads.cpp main)."""
    lib = load_host()
    o = default_options(**opts)
    st = SimuStats()
    err = C.create_string_buffer(4096)
    rc = lib.simu_run(config_path.encode(), C.byref(o), C.byref(st), err, len(err))
    if rc != 0:
        raise SimuError(f"simuReads failed (exit code {rc}): {err.value.decode(errors='replace')}")
    return st


class SimuTrainOptions(C.Structure):
    """simu_train_options (host/train.h): the options of the reference's seqToProfile + the additive ones"""
    _fields_ = [("bam", C.c_char_p), ("sam", C.c_char_p), ("target", C.c_char_p), ("vcf", C.c_char_p), ("ref", C.c_char_p),
                ("output", C.c_char_p), ("samtools", C.c_char_p), ("kmer", C.c_int32), ("bins", C.c_int32), ("device", C.c_int32),
                ("threads", C.c_int32), ("quiet", C.c_int32), ("stamp", C.c_char_p), ("max_reads", C.c_uint64)]


class SimuTrainStats(C.Structure):
    _fields_ = [("lines", C.c_uint64), ("reads_counted", C.c_uint64), ("gc_rejected", C.c_uint64), ("gc_windows", C.c_uint64),
                ("gc_pairs", C.c_uint64), ("skipped_overhang", C.c_uint64), ("sam_bytes", C.c_uint64), ("read_length", C.c_int32),
                ("bins", C.c_int32), ("gc_fitted", C.c_int32), ("capped", C.c_int32), ("t_reference", C.c_double), ("t_reads", C.c_double), ("t_total", C.c_double),
                ("insert_rate", C.c_double), ("del_rate", C.c_double), ("std_isize", C.c_double), ("gc_std", C.c_double)]


def train_profile(ref: str, vcf: str, output: str, sam: str = "", bam: str = "", target: str = "", samtools: str = "", kmer: int = 3,
                  bins: int = 50, device: int = 0, quiet: int = 1, stamp: str = None, max_reads: int = 0) -> SimuTrainStats:
    """`seqToProfile` in-process (src/seqToProfile.cpp main): reads (`sam`: a file of `samtools view` text, or `bam` through
    samtools as the reference does), the sample's VCF and the reference -> a .profile file; the per-read work runs on the GPU."""
    lib = load_host()
    lib.simu_train_default_options.argtypes = [C.POINTER(SimuTrainOptions)]
    lib.simu_train_default_options.restype = None
    lib.simu_train.argtypes = [C.POINTER(SimuTrainOptions), C.POINTER(SimuTrainStats), C.c_char_p, C.c_size_t]
    o = SimuTrainOptions()
    lib.simu_train_default_options(C.byref(o))
    o.bam, o.sam, o.target, o.vcf, o.ref, o.output, o.samtools = (x.encode() for x in (bam, sam, target, vcf, ref, output, samtools))
    o.kmer, o.bins, o.device, o.quiet, o.max_reads = kmer, bins, device, quiet, max_reads
    if stamp is not None:
        o.stamp = stamp.encode()
    st = SimuTrainStats()
    err = C.create_string_buffer(4096)
    rc = lib.simu_train(C.byref(o), C.byref(st), err, len(err))
    if rc != 0:
        raise SimuError(f"seqToProfile failed (exit code {rc}): {err.value.decode(errors='replace')}")
    return st


def release_cached_memory() -> None:
    """Device blocks that finished contexts left with the process go back to the runtime (sg_release_cached_memory)."""
    load_engine().sg_release_cached_memory()


class Session:
    """Step-by-step driver: inputs stay resident in HBM, the caller launches the sampling pass."""

    def __init__(self, config_path: str, **opts):
        self.lib = load_host()
        self.eng = load_engine()
        self._opts = default_options(**opts)
        self._h = C.c_void_p()
        self._err = C.create_string_buffer(4096)
        rc = self.lib.simu_open(config_path.encode(), C.byref(self._opts), C.byref(self._h), self._err, len(self._err))
        if rc != 0:
            raise SimuError(self._err.value.decode(errors="replace"))
        self.ctx = C.c_void_p(self.lib.simu_engine(self._h))

    def _check(self, rc):
        if rc != 0:
            raise SimuError(self._err.value.decode(errors="replace"))

    def _sg(self, rc, what):
        if rc != 0:
            raise SimuError(f"{what}: {self.eng.sg_last_error(self.ctx).decode(errors='replace')}")

    @property
    def planned_reads(self) -> int:
        return int(self.lib.simu_planned_reads(self._h))

    @property
    def n_chromosomes(self) -> int:
        return int(self.lib.simu_chromosome_count(self._h))

    def weighted_length(self, popu: int = 0) -> float:
        wl = C.c_double()
        self._check(self.lib.simu_weighted_length(self._h, popu, C.byref(wl), self._err, len(self._err)))
        return wl.value

    def set_reads(self, reads: int, popu: int = 0) -> None:
        self._check(self.lib.simu_set_reads(self._h, popu, int(reads), self._err, len(self._err)))

    def prepare_batch(self, chrom: int = 0, popu: int = 0) -> bool:
        hw = C.c_int()
        self._check(self.lib.simu_prepare_batch(self._h, popu, chrom, C.byref(hw), self._err, len(self._err)))
        return bool(hw.value)

    def set_stream(self, stream_handle: int) -> None:
        self._sg(self.eng.sg_set_stream(self.ctx, C.c_void_p(stream_handle)), "sg_set_stream")

    def set_seed(self, seed: int) -> None:
        self._sg(self.eng.sg_set_seed(self.ctx, int(seed)), "sg_set_seed")

    def sample(self) -> None:
        self._sg(self.eng.sg_sample(self.ctx), "sg_sample")

    def result(self):
        b1, b2, nf = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._sg(self.eng.sg_result(self.ctx, C.byref(b1), C.byref(b2), C.byref(nf)), "sg_result")
        return b1.value, b2.value, nf.value

    def kernel_times(self):
        ms = (C.c_float * 8)()
        self._sg(self.eng.sg_kernel_times(self.ctx, ms), "sg_kernel_times")
        return {n: float(ms[i]) for i, n in enumerate(SG_K_NAMES)}

    def compress(self):
        """BGZF-compress the FASTQ text of the last pass on the device; returns the compressed sizes."""
        g1, g2 = C.c_uint64(), C.c_uint64()
        self._sg(self.eng.sg_compress(self.ctx, C.byref(g1), C.byref(g2)), "sg_compress")
        return g1.value, g2.value

    def fetch_compressed(self, mate: int, nbytes: int, offset: int = 0) -> bytes:
        buf = C.create_string_buffer(max(nbytes, 1))
        self._sg(self.eng.sg_fetch_compressed(self.ctx, mate, offset, nbytes, buf), "sg_fetch_compressed")
        return buf.raw[:nbytes]

    def emit_info(self):
        """(items handed to the generic item code, whether the batch was re-emitted) of the last pass."""
        q, r = C.c_uint64(), C.c_int()
        self._sg(self.eng.sg_emit_info(self.ctx, C.byref(q), C.byref(r)), "sg_emit_info")
        return q.value, bool(r.value)

    def fetch(self, n1: int, n2: int):
        b1 = C.create_string_buffer(max(n1, 1))
        b2 = C.create_string_buffer(max(n2, 1))
        self._sg(self.eng.sg_fetch(self.ctx, b1, b2 if n2 else None), "sg_fetch")
        return b1.raw[:n1], b2.raw[:n2]

    def output_md5(self, chunk: int = 1 << 27):
        """md5 of each mate's whole FASTQ text of the last pass, fetched from the device in chunks."""
        import hashlib
        b1, b2, _ = self.result()
        out = []
        buf = C.create_string_buffer(chunk)
        for mate, total in ((0, b1), (1, b2)):
            if mate == 1 and not total:
                break
            h = hashlib.md5()
            for off in range(0, total, chunk):
                n = min(chunk, total - off)
                self._sg(self.eng.sg_fetch_range(self.ctx, mate, off, n, buf), "sg_fetch_range")
                h.update(memoryview(buf)[:n])
            out.append(h.hexdigest())
        return out

    # ---- detached output sets: the text of a pass drains to pinned host memory while the next pass is sampled ----
    def host_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._sg(self.eng.sg_host_alloc(self.ctx, nbytes, C.byref(p)), "sg_host_alloc")
        return p.value

    def host_free(self, ptr: int) -> None:
        self._sg(self.eng.sg_host_free(self.ctx, C.c_void_p(ptr)), "sg_host_free")

    def detach_outputs(self):
        h = C.c_void_p()
        self._sg(self.eng.sg_detach_outputs(self.ctx, C.byref(h)), "sg_detach_outputs")
        return h

    def outputs_sizes(self, h):
        tb, gb = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        self.eng.sg_outputs_sizes(h, tb, gb)
        return (tb[0], tb[1]), (gb[0], gb[1])

    def outputs_fetch_into(self, h, mate: int, compressed: bool, offset: int, nbytes: int, host_ptr: int) -> None:
        rc = self.eng.sg_outputs_fetch(h, mate, 1 if compressed else 0, offset, nbytes, C.cast(C.c_void_p(host_ptr), C.c_char_p))
        if rc != 0:
            raise SimuError("sg_outputs_fetch: " + self.eng.sg_outputs_last_error(h).decode(errors="replace"))

    def release_outputs(self, h) -> None:
        self._sg(self.eng.sg_release_outputs(self.ctx, h), "sg_release_outputs")

    def stats(self) -> SimuStats:
        st = SimuStats()
        self.lib.simu_get_stats(self._h, C.byref(st))
        return st

    def close(self):
        if self._h:
            self.lib.simu_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
