#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing, whole kernel and per basic block.

usage: isa_histogram.py listing.s <substring of the mangled kernel name> [--blocks] [--dump]
Classes: mfma, valu (v_* other than mfma / accvgpr), salu, lds (ds_*), vmem (global_/buffer_/scratch_), nop (s_nop), wait.
"""
import collections
import re
import sys


def kernel_body(path, key):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") is False and not re.match(r"^_Z\w+:", l):
            continue
        m = re.match(r"^(_Z\w+):", l)
        if m and key in m.group(1):
            start = i
            break
    if start is None:
        raise SystemExit(f"no kernel matching {key}")
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return lines[start], body


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    want_blocks = "--blocks" in sys.argv
    name, body = kernel_body(path, key)
    print(name)
    blocks, cur = [], ("entry", [])
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                blocks.append(cur)
                cur = (m.group(1), [])
            continue
        op = s.split()[0]
        nops = 0
        if op == "s_nop":
            nops = int(s.split()[1]) + 1
        cur[1].append((op, s, nops))
    blocks.append(cur)
    total = collections.Counter()
    ops = collections.Counter()
    for label, ins in blocks:
        c = collections.Counter()
        for op, s, nops in ins:
            c[classify(op)] += 1
            if nops:
                c["nop_states"] += nops
            ops[op] += 1
        total.update(c)
        if want_blocks and len(ins) >= 20:
            print(f"  {label:12s} n={len(ins):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("TOTAL " + " ".join(f"{k}={v}" for k, v in sorted(total.items())))
    if "--ops" in sys.argv:
        for op, n in ops.most_common(60):
            print(f"   {n:6d} {op}")
    if "--dump" in sys.argv:
        i = sys.argv.index("--dump")
        want = sys.argv[i + 1]
        for label, ins in blocks:
            if label == want:
                for op, s, _ in ins:
                    print("    " + s)


if __name__ == "__main__":
    main()
