#!/usr/bin/env python3
"""Static check of the SHIPPED gfx950 code for the hazard ADVICE round 4 names in the owner pass of the grid backward
(csrc/grid_kernels.hip, bucket_level_packed): its queue stream issues `global_load_dwordx3 ... nt` from inline asm into registers
`grp[][]` and awaits them with counted `s_waitcnt vmcnt(N)` asm statements.  The compiler's own wait-count insertion does not know
these loads exist, so nothing but the source's discipline stops it from COPYING, SPILLING or REUSING a destination register between the
issue and the wait (a loop back-edge rotation, a spill under register pressure, another TCNN_OWNER_GROUPS value): such an instruction
would read or clobber registers whose data is still in flight -- silently wrong gradients on the GPU, and no emulator test can see it
(the path is compiled out under TCNN_HOST_EMU).

The check: for every `global_load_dwordx3` in the kernels named on the command line (default: every k_grid_bucket_owner instance),
follow EVERY control-flow path from the load, counting the vector-memory operations issued after it, until an `s_waitcnt vmcnt(N)`
with N <= that count (vmcnt retires in issue order: at most N outstanding means this load has landed).  Any instruction on the way
that names one of the load's destination registers is a finding; so is a path that reaches the end of the kernel, or runs for more
than MAX_PATH instructions, without such a wait.

input: `llvm-objdump -d --no-show-raw-insn` of the gfx950 code object (llvm-objdump --offloading libtcnn_hip.so extracts it).
usage: check_asm_load_hazard.py disassembly.txt [kernel-name-fragment ...]; exit code 1 and one line per finding.
"""
import re
import sys

KERNEL = re.compile(r"^([0-9a-f]+) <(\S+)>:")
INST = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):")
TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
MAX_PATH = 4000
COUNT_CAP = 64


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse(path_or_lines):
    """-> {kernel: [(address, text, branch_target_address or None)]}"""
    lines = open(path_or_lines, errors="replace") if isinstance(path_or_lines, str) else path_or_lines
    kernels, cur, base = {}, None, 0
    for raw in lines:
        m = KERNEL.match(raw)
        if m:
            base = int(m.group(1), 16)
            cur = kernels.setdefault(m.group(2), [])
            continue
        if cur is None:
            continue
        m = INST.match(raw)
        if not m:
            continue
        text, addr = m.group(1).strip(), int(m.group(2), 16)
        target = None
        if text.startswith("s_cbranch") or text.startswith("s_branch"):
            t = TARGET.search(raw)
            if t:
                target = base + int(t.group(1), 16)
        cur.append((addr, text, target))
    return kernels


def check_kernel(name, insts, opcode="global_load_dwordx3"):
    index_of = {a: i for i, (a, _, _) in enumerate(insts)}
    findings = []
    n_loads = 0
    for i, (addr, text, _) in enumerate(insts):
        if not text.startswith(opcode):
            continue
        n_loads += 1
        dst = regs_of(text.split(",")[0])
        # depth-first over paths: (instruction index, vm ops issued since the load, instructions walked)
        stack, seen = [(i + 1, 0, 0)], set()
        while stack:
            j, count, walked = stack.pop()
            if (j, min(count, COUNT_CAP)) in seen:
                continue
            seen.add((j, min(count, COUNT_CAP)))
            if j >= len(insts) or walked > MAX_PATH:
                findings.append(f"{name}: load at {addr:#x} ({text}) reaches the end of the kernel / {MAX_PATH} instructions without a sufficient s_waitcnt vmcnt")
                break
            a, t, target = insts[j]
            if t.startswith("s_waitcnt"):
                m = VMCNT.search(t)
                if m and int(m.group(1)) <= count:
                    continue  # landed on this path
            elif t.startswith("s_endpgm"):
                continue  # the wave ends: nothing reads the registers any more
            elif regs_of(t) & dst:
                findings.append(f"{name}: `{t}` at {a:#x} touches v{sorted(regs_of(t) & dst)} while the load at {addr:#x} ({text}) may still be in flight "
                                f"({count} vector-memory operations issued since)")
                break
            nxt = count + (1 if VMEM.match(t) else 0)
            if t.startswith("s_branch"):
                if target in index_of:
                    stack.append((index_of[target], nxt, walked + 1))
                continue
            if t.startswith("s_cbranch") and target in index_of:
                stack.append((index_of[target], nxt, walked + 1))
            stack.append((j + 1, nxt, walked + 1))
    return findings, n_loads


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    fragments = argv[2:] or ["k_grid_bucket_owner"]
    bad, total = [], 0
    for name, insts in parse(argv[1]).items():
        if not any(f in name for f in fragments):
            continue
        f, n = check_kernel(name, insts)
        bad += f
        total += n
    for line in bad:
        print(line)
    print(f"{total} loads checked, {len(bad)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
