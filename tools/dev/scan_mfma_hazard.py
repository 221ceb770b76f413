#!/usr/bin/env python3
"""Scan gfx950 assembly for the hazards hipcc does NOT pad around an inline-asm MFMA (cdna_hip_programming.md, 'inline asm',
rule 2): the 16-bit kernels issue v_mfma_f32_16x16x32 through asm statements, so between such an MFMA and

  * a non-MFMA instruction that reads or writes one of its destination registers (XDL write -> VALU / VMEM / LDS access:
    up to 12 wait states for an 8-pass MFMA -- this scanner asks for WAIT_D), or
  * a VALU instruction that wrote one of its source registers just before it (VALU write -> XDL read: 2 wait states)

the required wait states must come from the instruction stream itself.  An MFMA that takes the destination whole as its
accumulator (the accumulate chain) needs none.  Wait states are counted along EVERY path of the control-flow graph (branch
targets and loop back-edges included: an MFMA at a loop tail is checked against the loop head): every instruction 1,
`s_nop N` N + 1; a path ends at s_endpgm.   usage: scan_mfma_hazard.py file.s"""
import re
import sys

WAIT_D = 12        # MFMA destination -> any other access (the 8-pass 16x16x32 forms: 11 + 1)
# the MFMAs the kernels issue through asm, and the states each needs behind it: conv_lpr.hip's fused first layer threads the 16-pass
# v_mfma_f32_32x32x2_f32 between the 16-bit ones (19 + 1)
SHAPES = {"v_mfma_f32_16x16x32": WAIT_D, "v_mfma_f32_32x32x2": 20}
WAIT_S = 2         # VALU write -> MFMA source
_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _instructions(path):
    """-> (instructions, label -> index of the first instruction behind it)"""
    ins, labels = [], {}
    for raw in open(path):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#") else ""
        if not line or line.startswith("//"):
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(ins)
            continue
        if line.startswith("."):
            continue
        ins.append(line)
    return ins, labels


def _successors(ins, labels, j):
    """indices control can reach right after instruction j (text order + branch targets, back-edges included)"""
    line = ins[j]
    if line.startswith("s_endpgm"):
        return []
    op = line.split()[0]
    if op == "s_branch" or op.startswith("s_cbranch"):
        tgt = labels.get(line.split()[-1])
        out = [] if tgt is None else [tgt]
        if op != "s_branch" and j + 1 < len(ins):
            out.append(j + 1)
        return out
    return [j + 1] if j + 1 < len(ins) else []


def _states(line):
    m = re.match(r"s_nop\s+(\d+)", line)
    return int(m.group(1)) + 1 if m else 1


def scan(path, shapes=None):
    shapes = SHAPES if shapes is None else {k: SHAPES.get(k, WAIT_D) for k in shapes}
    ins, labels = _instructions(path)
    preds = {}
    for j in range(len(ins)):
        for k in _successors(ins, labels, j):
            preds.setdefault(k, []).append(j)
    hits = []
    for i, line in enumerate(ins):
        if not line.startswith(tuple(shapes)):
            continue
        wait_d = next(v for k, v in shapes.items() if line.startswith(k))
        ops = [o.strip() for o in line.split(None, 1)[1].split(",")]
        dst, srcs = _regs(ops[0]), _regs(ops[1]) | _regs(ops[2])
        # --- destination hazard: every path out of the MFMA (a loop's back-edge continues at the loop head), until WAIT_D states
        best = {}                                          # instruction index -> fewest states it was reached with
        work = [(k, 0) for k in _successors(ins, labels, i)]
        found = None
        while work and found is None:
            j, states = work.pop()
            if states >= wait_d or best.get(j, wait_d) <= states:
                continue
            best[j] = states
            nxt = ins[j]
            if nxt.startswith("v_mfma"):
                nops = [o.strip() for o in nxt.split(None, 1)[1].split(",")]
                touched = (_regs(nops[1]) | _regs(nops[2])) & dst          # as A / B operand: a hazard; whole as C: the chain
                if touched or (_regs(nops[0]) & dst and _regs(nops[3]) != _regs(nops[0])):
                    found = (states, nxt)
                    break
            elif not nxt.startswith("s_") and _regs(nxt) & dst:
                found = (states, nxt)
                break
            for k in _successors(ins, labels, j):
                work.append((k, states + _states(nxt)))
        if found:
            hits.append(f"{path}: `{line}` then after {found[0]} states `{found[1]}`")
        # --- source hazard: a VALU write right in front of the MFMA, on any path into it
        work = [(k, 0) for k in preds.get(i, [])]
        seen = set()
        while work:
            j, states = work.pop()
            if states >= WAIT_S or (j, states) in seen:
                continue
            seen.add((j, states))
            prv = ins[j]
            if prv.startswith("v_") and not prv.startswith("v_mfma"):
                wr = _regs(prv.split(None, 1)[1].split(",")[0]) if " " in prv else set()
                if wr & (srcs | (_regs(ops[3]) if len(ops) > 3 else set())):
                    hits.append(f"{path}: `{prv}` {states} states before `{line}`")
                    break
            for k in preds.get(j, []):
                work.append((k, states + _states(prv)))
    return hits


if __name__ == "__main__":
    found = [h for p in sys.argv[1:] for h in scan(p)]
    print("\n".join(found) if found else "no unpadded MFMA hazard")
    sys.exit(1 if found else 0)
