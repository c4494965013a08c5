#!/usr/bin/env python3
"""Static issue-cost estimate of a kernel's basic blocks from hipcc's assembly output.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -o k.s file.hip
    python tools/isa_cost.py k.s <kernel-symbol-substring> [--min 20]

Every instruction gets the issue cost measured by tools/valu_microbench.hip at 4 waves per SIMD (cycles per wave
instruction per SIMD): 2.3 for the full-rate VALU subset (add, sub, and, or, xor, right shifts, mov), 2.8 for
v_bitop3_b32, 4.2 for every other VALU op (compares, selects, left shifts, bit-field ops, multiplies, three-operand
integer ops, lane reads/writes).  SALU / SMEM / LDS / VMEM instructions are counted, not priced (they issue beside
the VALU).  Output: one line per basic block with its loop depth (from the compiler's loop comments), instruction
counts by class and the VALU cycles of one pass through it."""
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32",
        "v_xnor_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}


def cost(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in FULL and not op.endswith("_sdwa"):
        return 2.3
    if base == "v_bitop3_b32":
        return 2.8
    if base in ("v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64"):
        return 4.2
    return 4.2


def main():
    path, sym = sys.argv[1], sys.argv[2]
    minc = float(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0.0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and sym in l.split(":")[0])
    blocks, cur = [], None
    depth = 0
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end") or t.startswith(".section"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", t)
        if m:
            d = re.search(r"Depth=(\d+)", t)
            if "Loop Header" in t or "Inner Loop Header" in t:
                depth = int(d.group(1)) if d else depth
            elif "in Loop" in t:
                depth = int(d.group(1)) if d else depth
            else:
                depth = 0
            cur = {"name": m.group(1), "depth": depth, "valu": 0, "cyc": 0.0, "salu": 0, "smem": 0, "lds": 0, "vmem": 0, "branch": 0,
                   "ops": {}}
            blocks.append(cur)
            continue
        if cur is None:
            cur = {"name": "entry", "depth": 0, "valu": 0, "cyc": 0.0, "salu": 0, "smem": 0, "lds": 0, "vmem": 0, "branch": 0, "ops": {}}
            blocks.append(cur)
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["cyc"] += cost(op)
            key = re.sub(r"_(e32|e64)$", "", op)
            cur["ops"][key] = cur["ops"].get(key, 0) + 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            cur["branch"] += 1
            cur["salu"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            cur["smem"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            cur["vmem"] += 1
    tot = {"valu": 0, "cyc": 0.0, "salu": 0, "lds": 0, "vmem": 0}
    print(f"{'block':12s} depth  VALU  cycles  SALU SMEM  LDS VMEM   top VALU ops")
    for b in blocks:
        for k in tot:
            tot[k] += b[k]
        if b["cyc"] < minc and b["salu"] < minc:
            continue
        top = ", ".join(f"{k}x{v}" for k, v in sorted(b["ops"].items(), key=lambda kv: -kv[1])[:6])
        print(f"{b['name']:12s} {b['depth']:5d} {b['valu']:5d} {b['cyc']:7.0f} {b['salu']:5d} {b['smem']:4d} {b['lds']:4d} {b['vmem']:4d}   {top}")
    print(f"total: {tot['valu']} VALU = {tot['cyc']:.0f} cycles, {tot['salu']} SALU, {tot['lds']} LDS, {tot['vmem']} VMEM")


if __name__ == "__main__":
    main()
