#!/usr/bin/env python3
"""Static check of the compiled gfx950 ISA for the hazard behind round 2's "spilled kernel gives varying results" finding
(profiles/r03_mfma_branch_hazard.txt): an MFMA result read too soon after a TAKEN BRANCH.

Within straight-line code hipcc pads the wait states an MFMA result needs before anything but a chained MFMA (same registers
as C) touches it.  Across a taken branch it under-pads: in the experiment kernel the path

        v_mfma_f32_16x16x32_f16 v[172:175], ..., v[172:175]
        s_cbranch_vccnz .LBB5_43
    .LBB5_43:
        s_nop 6
        v_max_f32_e32 v122, v172, v172          <- 8 wait states after the MFMA

is what hipcc emits where it sees the hazard: 8 wait states (the branch counts as one) for a 4-pass MFMA on gfx950 (LLVM:
passes + 3 + 1; 12 for the 8-pass shapes).  In the experiment kernel's SECOND such path the successor block starts with
`v_max_f32_e32 v1, v162, v162` and no s_nop at all -- 1 wait state after `v_mfma_f32_16x16x32_f16 v[162:165], ...`: stale v162 on
the GPU (wrong outputs that changed from run to run); two more wait states in front of the branch cured it.  This script demands
the compiler's own figure -- WAIT_16 = 8 states, WAIT_32 = 12 for the 32x32 shapes -- on every path that leaves an MFMA through
a taken branch.

usage: check_mfma_branch_hazard.py file.s [...]     (device assembly from `hipcc -save-temps`, or llvm-objdump -d output)
exit code 1 and one line per finding if a path is short.
"""
import re
import sys

WAIT_16 = 8   # 4-pass MFMA result -> any reader but the chained MFMA (gfx950)
WAIT_32 = 12  # 8-pass
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
KERNEL = re.compile(r"^(_Z\w+):")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
BRANCH = re.compile(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse(path):
    """-> {kernel: [(kind, payload)]} with kind in label / inst"""
    kernels, cur = {}, None
    for raw in open(path, errors="replace"):
        line = raw.rstrip("\n")
        code = line.split(";")[0].rstrip()
        m = KERNEL.match(code)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if code.startswith(".Lfunc_end"):
            cur = None
            continue
        m = LABEL.match(code)
        if m:
            cur.append(("label", m.group(1)))
            continue
        if code.startswith("\t") and not code.strip().startswith("."):
            cur.append(("inst", code.strip()))
    return kernels


def states(inst):
    m = re.match(r"s_nop\s+(\d+)", inst)
    return int(m.group(1)) + 1 if m else 1


def operands(inst):
    parts = inst.split(None, 1)
    return [p.strip() for p in parts[1].split(",")] if len(parts) > 1 else []


def check_kernel(name, items):
    findings = []
    label_at = {p: i for i, (k, p) in enumerate(items) if k == "label"}
    for i, (kind, inst) in enumerate(items):
        if kind != "inst":
            continue
        b = BRANCH.match("\t" + inst)
        if not b or b.group(2) not in label_at:
            continue
        # MFMAs in the straight-line code before this branch, with the wait states already spent after them (the branch counts as 1)
        pending = []  # (dst regs, states still needed at the target, mfma text)
        spent = 1
        j = i - 1
        while j >= 0 and spent < WAIT_32:
            k, t = items[j]
            if k == "inst":
                if t.startswith("s_branch") or t.startswith("s_endpgm") or t.startswith("s_setpc"):
                    break
                if t.startswith("v_mfma") or t.startswith("v_smfmac"):
                    need = (WAIT_32 if "32x32" in t else WAIT_16) - spent
                    if need > 0:
                        pending.append((regs_of(operands(t)[0]), need, t))
                spent += states(t)
            j -= 1
        if not pending:
            continue
        # the target block: who touches those registers within the missing wait states?
        for dst, need, mfma in pending:
            spent_t = 0
            j = label_at[b.group(2)] + 1
            while j < len(items) and spent_t < need:
                k, t = items[j]
                if k == "inst":
                    if t.startswith("s_branch") or t.startswith("s_endpgm"):
                        break
                    ops = operands(t)
                    touched = regs_of(" ".join(ops))
                    if t.startswith("v_mfma") and len(ops) == 4 and regs_of(ops[3]) == dst and regs_of(ops[0]) == dst and not (regs_of(ops[1] + " " + ops[2]) & dst):
                        touched = set()  # accumulate chain: the one consumer that needs no wait states
                    # a memory LOAD into the registers (the MFMA's result is dead on this path) lands tens of cycles later at the earliest
                    if re.match(r"(ds_read|ds_load|global_load|scratch_load|buffer_load|flat_load)", t) and not (regs_of(" ".join(ops[1:])) & dst):
                        touched = set()
                    if touched & dst:
                        findings.append(f"{name}: `{mfma}` -> `{inst}` -> `{t}` after {(WAIT_32 if '32x32' in mfma else WAIT_16) - need + spent_t} wait states")
                        break
                    spent_t += states(t)
                j += 1
    return findings


def main(paths):
    bad = []
    n_kernels = n_mfma = 0
    for p in paths:
        for name, items in parse(p).items():
            n_kernels += 1
            n_mfma += sum(1 for k, t in items if k == "inst" and t.startswith("v_mfma"))
            bad += check_kernel(name, items)
    for f in bad:
        print("HAZARD", f)
    print(f"checked {n_kernels} kernels, {n_mfma} MFMA instructions: {len(bad)} short MFMA -> taken branch -> reader path(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
