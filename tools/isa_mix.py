#!/usr/bin/env python3
"""tools/isa_mix.py -- static instruction mix of a kernel's hot loop, priced in issue cycles.

Input: the compiler's assembly of a kernel source (hipcc -S --cuda-device-only with the build's flags: same compiler, same flags, same code as
the shipped object).  The kernel's outermost loop (the symbol loop of ofdm_demap_kernel) is cut into basic blocks; blocks are listed with their
instruction counts per class, and the classes are priced with the issue costs bench.py uses:

    simple   2 cycles  32-bit VOP1 / VOP2 encodings of add / sub / mul / fmac / and / or / xor / mov / shifts (MI355X_MICROARCH.md; tools/ubench/valu_rates.hip)
    vop3     4 cycles  every VOP3 / VOP3P(int) / DPP / SDWA form, compares, conversions, min / max, bfe / bfi / perm, add3 ...
    pk_f32   5 cycles  v_pk_fma / add / mul_f32 (two lanes' worth per lane: measured 5.0 .. 6.0 clocks against 2.4 .. 2.9 of the plain forms)
    trans    8 cycles  v_sqrt / rcp / rsq / exp / log / sin / cos
    swap     8 cycles  v_permlane16_swap / v_permlane32_swap (measured 8.3)

`--weights L1=w,L2=w,...` gives blocks a weight other than 1: 0 for blocks off the steady-state path (the per-sample stale-tail path, list
appends), 0.5 for blocks only two of the four waves of a workgroup enter (the byte pack by 96 threads), and `--remainder L1,L2,...` names the
blocks whose mix prices whatever the measured instruction count per unit (PMC, bench.py) exceeds the weighted static count by (the parity
guard's per-bin repeat, entered by a wave whenever one of its lanes trips the per-thread threshold; prologue).  Output: text table + one JSON
line (the numbers bench.py reads).
"""
import argparse
import json
import re
import subprocess
import sys

COST = {"simple": 2.0, "vop3": 4.0, "pk_f32": 5.0, "trans": 8.0, "swap": 8.0}
SIMPLE = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
          "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32", "v_mul_u32_u24", "v_mul_i32_i24"}
TRANS = ("v_sqrt", "v_rcp", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")


def classify(op, operands):
    if op.startswith("v_permlane") and "swap" in op:
        return "swap"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "pk_f32"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith(("_e64", "_dpp", "_sdwa")) or "row_" in operands or "quad_perm" in operands or "sel:" in operands:
        return "vop3"
    if base in SIMPLE and not re.search(r"\b(abs|neg|clamp|mul:|div:)\b|\|", operands):
        # (v_fma_f32 has no 32-bit encoding but issues at the plain rate: measured 2.9 against 2.6 .. 2.7 of v_add / v_mul_f32)
        return "simple"
    return "vop3"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", required=True, help="assembly text (hipcc -S --cuda-device-only ...)")
    ap.add_argument("--kernel", required=True, help="substring of the mangled kernel name")
    ap.add_argument("--weights", default="", help="label=weight,... (default weight 1)")
    ap.add_argument("--remainder", default="", help="labels whose mix prices the instructions the PMC count has beyond the weighted static count")
    ap.add_argument("--source-sha", default="", help="recorded in the JSON: sha256 of the kernel's sources")
    ap.add_argument("--per-iteration", type=float, default=1.0, help="loop iterations per unit (e.g. 0.5: the loop body handles two symbols)")
    args = ap.parse_args()
    lines = open(args.asm).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(args.kernel), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    # the outermost loop with the most instructions: header label = the one most blocks name as "Loop Header" / "in Loop: Header=..."
    hdr_count = {}
    for l in body:
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=1", l)
        if m:
            hdr_count[m.group(1)] = hdr_count.get(m.group(1), 0) + 1
    hdr = max(hdr_count, key=hdr_count.get)
    weights = {kv.split("=")[0].strip(): float(kv.split("=")[1]) for kv in args.weights.split(",") if "=" in kv}
    remainder = {c.strip() for c in args.remainder.split(",") if c.strip()}
    blocks, cur, inloop = [], None, False
    for l in body:
        m = re.match(r"^\.L(BB\d+_\d+):(.*)", l)
        if m:
            label, rest = m.group(1), m.group(2)
            inloop = label == hdr or ("Header=%s " % hdr) in rest or ("Parent Loop %s " % hdr) in rest or ("Parent Loop %s\t" % hdr) in rest
            cur = {"label": label, "counts": {}, "other": {}, "branch": ""} if inloop else None
            if cur:
                blocks.append(cur)
            continue
        if cur is None or not l.startswith("\t") or l.strip().startswith((";", ".")):
            if cur is not None and "Parent Loop" in l:
                pass
            continue
        parts = l.strip().split(None, 1)
        op, operands = parts[0], parts[1] if len(parts) > 1 else ""
        if op.startswith("v_"):
            k = classify(op, operands)
            cur["counts"][k] = cur["counts"].get(k, 0) + 1
        else:
            k = "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "salu" if op.startswith("s_") else "other"
            cur["other"][k] = cur["other"].get(k, 0) + 1
            if op.startswith("s_cbranch") or op == "s_branch":
                cur["branch"] += " %s %s" % (op, operands)
    total, tot_other, rem = {}, {}, {}
    print("loop %s of %s: %d blocks" % (hdr, args.kernel, len(blocks)))
    for b in blocks:
        w = weights.get(b["label"], 1.0)
        valu = sum(b["counts"].values())
        print("  %-10s w %.2f%s valu %4d %-62s other %-42s%s" % (b["label"], w, " R" if b["label"] in remainder else "  ", valu, json.dumps(b["counts"], sort_keys=True),
                                                              json.dumps(b["other"], sort_keys=True), b["branch"]))
        for k, v in b["counts"].items():
            total[k] = total.get(k, 0) + w * v
            if b["label"] in remainder:
                rem[k] = rem.get(k, 0) + v
        for k, v in b["other"].items():
            tot_other[k] = tot_other.get(k, 0) + w * v
    s = args.per_iteration
    valu = sum(total.values()) * s
    cycles = sum(COST[k] * v for k, v in total.items()) * s
    rem_n = sum(rem.values())
    out = {"kernel": args.kernel, "loop": hdr, "weights": weights, "remainder_blocks": sorted(remainder), "source_sha256": args.source_sha,
           "per_unit": {k: v * s for k, v in sorted(total.items())}, "other_per_unit": {k: v * s for k, v in sorted(tot_other.items())},
           "valu_per_unit": valu, "issue_cycles_per_unit": cycles, "mean_cycles_per_valu": cycles / valu if valu else None,
           "remainder_cycles_per_valu": (sum(COST[k] * v for k, v in rem.items()) / rem_n) if rem_n else None, "cost_model": COST}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
