"""Degenerate configurations around two parity cases (shared by test_oracle_edge_inputs_vs_reference.py and
test_gpu_edge_inputs.py): what the reference does with them -- mostly quietly -- the oracle and the GPU path must do too."""
import os
import re

import cases


def _edit(text, **kv):
    for k, v in kv.items():
        if re.search(rf"^{k} = .*$", text, re.M):
            text = re.sub(rf"^{k} = .*$", f"{k} = {v}", text, flags=re.M)
        else:
            text += f"{k} = {v}\n"
    return text


def _bed_variants(bed):
    five = bed[:5]
    return {
        "bed_empty": [],
        "bed_absent_chromosome": five + ["chrZ\t100\t900"],
        "bed_beyond_contig_end": five + ["chr21\t899900\t900500"],
        "bed_start_after_end": five + ["chr21\t5000\t4000"],
        "bed_zero_length": five + ["chr21\t5000\t5000"],
        "bed_at_position_0": ["chr20\t0\t400"] + five,
        "bed_rows_twice": five + five,
        "bed_one_ten_base_target": ["chr20\t40000\t40010"],
        "bed_names_without_chr": [r.replace("chr", "") for r in bed[:6]],
    }


WGS_VARIANTS = {
    "coverage_0": dict(coverage=0), "coverage_fraction_below_1": dict(coverage=0.01), "coverage_3_7": dict(coverage=3.7),
    "insert_below_read_length": dict(insertSize=60), "insert_equal_read_length": dict(insertSize=74), "insert_5000": dict(insertSize=5000),
    "ploidy_4": dict(ploidy=4), "ploidy_1_with_cn_gain": dict(ploidy=1), "layout_se": dict(layout="SE"), "threads_0": dict(threads=0),
}


def _variation_variants(rows):
    popu, chrom = rows[0].split("\t")[1:3]
    return {
        "variation_file_empty": [],
        "variation_insertion_at_1": rows + [f"i\t{popu}\t{chrom}\t1\tACGT\thomo"],
        "variation_insertion_past_the_end": rows + [f"i\t{popu}\t{chrom}\t1000000000\tACGT\thomo"],
        "variation_deletion_past_the_end": rows + [f"d\t{popu}\t{chrom}\t999999990\t50\thomo"],
        "variation_on_absent_chromosome": rows + [f"i\t{popu}\tchrQ\t5000\tACGT\thomo"],
        "variation_rows_twice": rows + rows,
        "variation_lower_case_insertion": rows + [f"i\t{popu}\t{chrom}\t7000\tacgtn\thet"],
        "variation_unknown_population": rows + [f"i\tnobody\t{chrom}\t5000\tACGT\thomo"],
        "variation_short_row": rows + [f"s\t{popu}\t{chrom}\t5000\tA\thomo"],
    }


ABUNDANCE_VARIANTS = {   # rows of the abundance file of the four-population mixture case (one mixture per row)
    "abundance_sum_below_1": ["1.0\t0\t0\t0", "0.1\t0.1\t0.1\t0.1"],
    "abundance_sum_above_1": ["1.0\t0\t0\t0", "0.5\t0.5\t0.5\t0.5"],
    "abundance_zero_entry": ["1.0\t0\t0\t0", "0.5\t0\t0.25\t0.25"],
    "abundance_three_columns": ["1.0\t0\t0\t0", "0.5\t0.25\t0.25"],
    "abundance_five_columns": ["1.0\t0\t0\t0", "0.2\t0.2\t0.2\t0.2\t0.2"],
    "abundance_negative_entry": ["1.0\t0\t0\t0", "0.7\t-0.1\t0.2\t0.2"],
    "abundance_one_row": ["1.0\t0\t0\t0"],
    "abundance_three_rows": ["1.0\t0\t0\t0", "0.3\t0.25\t0.35\t0.1", "0.25\t0.25\t0.25\t0.25"],
}


def literal_x(seq, other=False):
    """`seq` with 400 short runs of X / x (and a few R) put in; other=True: the same places hold R / r instead of X / x
    (what --strict-bases makes of an X: an unknown base like any other)."""
    import random
    rng = random.Random(5)
    out = bytearray(seq)
    for _ in range(400):
        p = rng.randrange(10, len(out) - 10)
        n = rng.choice([1, 1, 2, 2, 3, 5])
        ch = rng.choice([b"X", b"X", b"x", b"R"])
        if other:
            ch = {b"X": b"R", b"x": b"r"}.get(ch, ch)
        out[p:p + n] = ch * n
    return bytes(out)


def _fasta_variants():
    """FASTA files fastahack indexes (Fasta.cpp:45-85) in shapes the ingest kernels have to cope with."""
    from simuscop_amd import synth
    a = synth.synth_contig(30000, 77, 0, n_runs=False).tobytes()
    b = synth.synth_contig(21000, 77, 1, n_runs=False).tobytes()

    def wrap(seq, w, eol=b"\n"):
        return b"".join(seq[i:i + w] + eol for i in range(0, len(seq), w))

    return {
        # a literal X walks the reference's k-mer trie as the place holder of the short contexts (Profile.cpp:94-101): runs of
        # one to five, lower case too, next to other non-ACGT characters
        "fasta_literal_x": [b">chr1\n", wrap(literal_x(a), 60), b">chr2\n", wrap(b, 60)],
        # a name met again: listed twice, both resolved to the FIRST sequence (Fasta.cpp:67,84-97,198)
        "fasta_repeated_name": [b">chr1\n", wrap(a, 60), b">chr2\n", wrap(b, 60), b">chr1\n", wrap(a[:12000][::-1], 60)],
        "fasta_lf": [b">chr1\n", wrap(a, 60), b">chr2\n", wrap(b, 60)],
        "fasta_crlf": [b">chr1\r\n", wrap(a, 60, b"\r\n"), b">chr2\r\n", wrap(b, 60, b"\r\n")],
        "fasta_header_with_description": [b">chr1 some description here\n", wrap(a, 60), b">chr2 other\n", wrap(b, 60)],
        "fasta_one_line_per_contig": [b">chr1\n", a + b"\n", b">chr2\n", b + b"\n"],
        "fasta_no_final_newline": [b">chr1\n", wrap(a, 60), b">chr2\n", wrap(b, 60)[:-1]],
        "fasta_blank_line_between_contigs": [b">chr1\n", wrap(a, 60), b"\n", b">chr2\n", wrap(b, 60)],
        "fasta_lower_case": [b">chr1\n", wrap(a.lower(), 60), b">chr2\n", wrap(b, 60)],
        "fasta_empty_contig": [b">chr1\n", wrap(a, 60), b">chr3\n", b">chr2\n", wrap(b, 60)],
        "fasta_trailing_space": [b">chr1\n", wrap(a, 60, b" \n"), b">chr2\n", wrap(b, 60)],
    }


_FASTA_NAMES = ["fasta_blank_line_between_contigs", "fasta_crlf", "fasta_lf", "fasta_empty_contig", "fasta_header_with_description", "fasta_lower_case",
                "fasta_no_final_newline", "fasta_one_line_per_contig", "fasta_trailing_space", "fasta_literal_x", "fasta_repeated_name"]
_VARIATION_NAMES = sorted(_variation_variants(["i\tp\tc\t1\tA\thomo"]))
NAMES = sorted(WGS_VARIANTS) + sorted(_bed_variants(["chr20\t1\t2"] * 6)) + _VARIATION_NAMES + ["snp_on_absent_chromosome"] + sorted(ABUNDANCE_VARIANTS) + _FASTA_NAMES


def build(name, workdir):
    """Write the inputs of variant `name` under workdir; returns the config path (its output directory: workdir/out)."""
    os.makedirs(workdir, exist_ok=True)
    if name in WGS_VARIANTS:
        base = cases.build_case("wgs_pe_variants", os.path.join(workdir, "base"))
        text = _edit(open(base).read(), **WGS_VARIANTS[name])
    elif name in _FASTA_NAMES:
        fa = os.path.join(workdir, "ref.fa")
        with open(fa, "wb") as f:
            f.write(b"".join(_fasta_variants()[name]))
        text = (f"ref = {fa}\nprofile = {os.path.join(cases.TESTDATA, cases.PROFILES['xten'])}\nname = fa\noutput = x\nlayout = PE\n"
                f"threads = 1\nverbose = 0\ncoverage = 3\ninsertSize = 350\n")
    elif name in ABUNDANCE_VARIANTS:
        base = cases.build_case("tumor_se_mixture", os.path.join(workdir, "base"))
        ap = os.path.join(workdir, "abundance.txt")
        with open(ap, "w") as f:
            f.write("\n".join(ABUNDANCE_VARIANTS[name]) + "\n")
        text = _edit(open(base).read(), abundance=ap)
    elif name in _VARIATION_NAMES or name == "snp_on_absent_chromosome":
        base = cases.build_case("wgs_pe_variants", os.path.join(workdir, "base"))
        text = open(base).read()
        if name == "snp_on_absent_chromosome":
            sp = os.path.join(workdir, "snp.txt")
            with open(sp, "w") as f:
                f.write("chrQ\t100\trs1\tA\tG\t0.5\n")
            text = _edit(text, snp=sp)
        else:
            rows = _variation_variants(open(os.path.join(workdir, "base", "variations.txt")).read().splitlines())[name]
            vp = os.path.join(workdir, "variations.txt")
            with open(vp, "w") as f:
                f.write("\n".join(rows) + ("\n" if rows else ""))
            text = _edit(text, variation=vp)
    else:
        base = cases.build_case("wes_pe_targets", os.path.join(workdir, "base"))
        bed = open(os.path.join(workdir, "base", "targets.bed")).read().splitlines()
        rows = _bed_variants(bed)[name]
        bp = os.path.join(workdir, "targets.bed")
        with open(bp, "w") as f:
            f.write("\n".join(rows) + ("\n" if rows else ""))
        text = _edit(open(base).read(), target=bp, coverage=10)
    text = _edit(text, output=os.path.join(workdir, "out"))
    cfg = os.path.join(workdir, "config.txt")
    with open(cfg, "w") as f:
        f.write(text)
    return cfg


# Every name runs through the GPU path too.  Three reference quirks were the product's own reading in round 3 (CR LF line
# ends read like LF, a literal X an unknown base, a repeated contig name keeping its last sequence); the default is now the
# reference's behaviour -- "fasta_crlf", "fasta_literal_x", "fasta_repeated_name" are byte-parity cases like the others --
# and the old readings sit behind --crlf-as-lf / --strict-bases / --unique-contigs (tests/test_gpu_edge_inputs.py).
GPU_NAMES = list(NAMES)
