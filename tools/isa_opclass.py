#!/usr/bin/env python3
"""Opcode-class table of one loop of a kernel from hipcc's --save-temps ISA (where do the VALU
instructions of the t-score tile go?).

    python tools/isa_opclass.py <file.s> <kernel-symbol-substring> [marker-opcode [min-count]]

Finds the innermost loop (label ... backward branch) of the kernel that contains `marker-opcode`
(default v_rsq_f64 -- the t-score tile is the only loop with a float64 reciprocal square root) and prints the
static instruction count per class for that loop body, plus the whole kernel for reference.  Static counts of a
loop body = dynamic counts per iteration (one tile) as long as the body has no inner loop that iterates, which
holds for the tile (its rare plateau walk is a separate, normally skipped block: reported on its own)."""
import re
import sys
from collections import Counter, OrderedDict

CLASSES = OrderedDict([
    ("fp64 arith (add/mul/fma)", r"^v_(add|mul|fma)_f64"),
    ("fp64 rcp/rsq seeds", r"^v_(rcp|rsq)_f64"),
    ("fp64 compare", r"^v_cmp\w*_f64|^v_cmpx\w*_f64"),
    ("cvt f32->f64", r"^v_cvt_f64_f32"),
    ("clip (v_med3_f32)", r"^v_med3_f32"),
    ("select (v_cndmask)", r"^v_cndmask"),
    ("DPP / lane moves", r"_dpp|^v_readlane|^v_readfirstlane|^v_writelane|^v_permlane|^v_mov_b32.*row_|^v_mov_b64.*row_"),
    ("int compare", r"^v_cmp"),
    ("int / address / bit ops", r"^v_(add|sub|lshl|lshr|ashr|and|or|xor|mad|mul_lo|mul_u|bfe|bfi|min_u|max_u|min_i|max_i|add3|lshl_add|lshl_or|and_or|or3|not|bcnt|mbcnt|perm|alignbit|cvt_u|cvt_i|cvt_f32_u|add_co|addc|subrev|sub_co|subb)"),
    ("moves (v_mov / v_accvgpr)", r"^v_mov|^v_accvgpr|^v_pk_mov"),
    ("other VALU", r"^v_"),
    ("LDS read", r"^ds_read|^ds_load"),
    ("LDS write", r"^ds_write|^ds_store"),
    ("LDS atomic / bpermute", r"^ds_(add|bpermute|permute|swizzle|max|min|or|and)"),
    ("global/flat memory", r"^(global|flat|buffer|scratch)_"),
    ("scalar ALU / moves", r"^s_(?!waitcnt|barrier|cbranch|branch|nop|endpgm|load|sleep|setprio)"),
    ("scalar memory", r"^s_load"),
    ("branches", r"^s_(c?branch)"),
    ("waitcnt", r"^s_waitcnt"),
    ("barrier", r"^s_barrier"),
    ("nop/other", r"."),
])


def classify(op):
    for name, pat in CLASSES.items():
        if re.search(pat, op):
            return name
    return "nop/other"


def kernel_body(lines, sym):
    start = None
    for i, ln in enumerate(lines):
        if start is None and re.match(r"^[_A-Za-z0-9.$]*%s[_A-Za-z0-9.$]*:" % re.escape(sym), ln):
            start = i
        elif start is not None and ln.startswith(".Lfunc_end"):  # (early exits put s_endpgm in mid-body)
            return lines[start:i]
    raise SystemExit("kernel not found")


def main():
    path, sym = sys.argv[1], sys.argv[2]
    marker = sys.argv[3] if len(sys.argv) > 3 else "v_rsq_f64"
    body = kernel_body(open(path).read().splitlines(), sym)
    label_at, instrs = {}, []
    for ln in body:
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                label_at[s[:-1]] = len(instrs)
            continue
        if s.endswith(":"):
            label_at[s[:-1]] = len(instrs)
            continue
        instrs.append(s)
    ops = [i.split()[0] + (" " + " ".join(i.split()[1:]) if "dpp" in i or "row_" in i else "") for i in instrs]
    # backward branches -> loops (target index <= branch index)
    loops = []
    for k, ins in enumerate(instrs):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= k:
            loops.append((label_at[m.group(1)], k))
    need = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # occurrences of the marker the loop must hold
    cand = [(a, b) for a, b in loops if sum(marker in instrs[j] for j in range(a, b + 1)) >= need]
    if not cand:
        raise SystemExit("no loop with marker " + marker)
    a, b = min(cand, key=lambda t: t[1] - t[0])
    inner = [(x, y) for x, y in loops if a < x and y < b]

    def table(rng, title):
        c = Counter(classify(ops[j]) for j in rng)
        valu = sum(v for k_, v in c.items() if k_ in list(CLASSES)[:11])
        print(f"## {title}: {len(list(rng))} instructions, {valu} VALU")
        for name in CLASSES:
            if c.get(name):
                print(f"  {name:34s} {c[name]:6d}")
        return c

    excl = set()
    for x, y in inner:
        excl.update(range(x, y + 1))
    table([j for j in range(a, b + 1) if j not in excl], f"loop body with {marker} (inner loops excluded)")
    for x, y in inner:
        table(range(x, y + 1), "  inner loop (e.g. plateau walk), per iteration")
    table(range(len(instrs)), "whole kernel (static)")


if __name__ == "__main__":
    main()
