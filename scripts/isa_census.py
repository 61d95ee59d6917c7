#!/usr/bin/env python3
"""Loop-level census of a gfx950 assembly listing (hipcc -save-temps .s): for every backward branch, the
instruction mix of the span label..branch (innermost spans first).  Usage: isa_census.py file.s [min_valu]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")):
        return "fp64"
    if op.startswith("v_mov_b64") or op.startswith("v_mov_b32") or op.startswith("v_accvgpr") or op.startswith("v_pk_mov"):
        return "mov"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "vother"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    min_valu = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lines = open(path).read().split("\n")
    labels = {}
    ins = []          # (lineno, op, text)
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", ln)
        if m and not ln.strip().startswith((".", ";")):
            ins.append((i + 1, m.group(1), m.group(2)))
    loops = []
    for idx, (ln, op, txt) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = txt.split()[0].strip()
            if tgt in labels and labels[tgt] <= idx:
                loops.append((labels[tgt], idx, tgt))
    loops.sort(key=lambda t: t[1] - t[0])
    print("%-12s %7s %6s | %5s %5s %5s %5s %6s | %4s %4s %4s %4s %4s" %
          ("label", "line", "instr", "fp64", "mov", "sel", "lane", "vother", "lds", "smem", "vmem", "salu", "br"))
    for a, b, tgt in loops:
        c = Counter(classify(op) for _, op, _ in ins[a:b + 1])
        valu = c["fp64"] + c["mov"] + c["sel"] + c["lane"] + c["vother"]
        if valu < min_valu:
            continue
        print("%-12s %7d %6d | %5d %5d %5d %5d %6d | %4d %4d %4d %4d %4d" %
              (tgt, ins[a][0], b - a + 1, c["fp64"], c["mov"], c["sel"], c["lane"], c["vother"], c["lds"], c["smem"],
               c["vmem"], c["salu"], c["br"]))


if __name__ == "__main__":
    main()
