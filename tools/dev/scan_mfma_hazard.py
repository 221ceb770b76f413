#!/usr/bin/env python3
"""Scan gfx950 assembly for the hazards hipcc does NOT pad around an inline-asm MFMA (cdna_hip_programming.md, 'inline asm',
rule 2): the 16-bit kernels issue v_mfma_f32_16x16x32 through asm statements, so between such an MFMA and

  * a non-MFMA instruction that reads or writes one of its destination registers (XDL write -> VALU / VMEM / LDS access:
    up to 12 wait states for an 8-pass MFMA -- this scanner asks for WAIT_D), or
  * a VALU instruction that wrote one of its source registers just before it (VALU write -> XDL read: 2 wait states)

the required wait states must come from the instruction stream itself.  An MFMA that takes the destination whole as its
accumulator (the accumulate chain) needs none.  Wait states are counted conservatively in text order across labels: every
instruction 1, `s_nop N` N + 1; the look-ahead stops at s_endpgm.   usage: scan_mfma_hazard.py file.s"""
import re
import sys

WAIT_D = 12        # MFMA destination -> any other access
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
    ins = []
    for raw in open(path):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#") else ""
        if not line or line.startswith(".") or line.endswith(":") or line.startswith("//"):
            continue
        ins.append(line)
    return ins


def scan(path):
    ins = _instructions(path)
    hits = []
    for i, line in enumerate(ins):
        if not line.startswith("v_mfma_f32_16x16x32"):
            continue
        ops = [o.strip() for o in line.split(None, 1)[1].split(",")]
        dst, srcs = _regs(ops[0]), _regs(ops[1]) | _regs(ops[2])
        # --- destination hazard
        states = 0
        for j in range(i + 1, min(i + 1 + 4 * WAIT_D, len(ins))):
            nxt = ins[j]
            if nxt.startswith("s_endpgm") or states >= WAIT_D:
                break
            if nxt.startswith("v_mfma"):
                nops = [o.strip() for o in nxt.split(None, 1)[1].split(",")]
                touched = (_regs(nops[1]) | _regs(nops[2])) & dst          # as A / B operand: a hazard; whole as C: the chain
                if touched or (_regs(nops[0]) & dst and _regs(nops[3]) != _regs(nops[0])):
                    hits.append(f"{path}: `{line}` then after {states} states `{nxt}`")
                    break
            elif _regs(nxt) & dst:
                hits.append(f"{path}: `{line}` then after {states} states `{nxt}`")
                break
            m = re.match(r"s_nop\s+(\d+)", nxt)
            states += int(m.group(1)) + 1 if m else 1
        # --- source hazard: a VALU write right in front of the MFMA
        states = 0
        for j in range(i - 1, max(i - 1 - WAIT_S, -1), -1):
            prv = ins[j]
            m = re.match(r"s_nop\s+(\d+)", prv)
            if m:
                states += int(m.group(1)) + 1
            else:
                if prv.startswith("v_") and not prv.startswith("v_mfma"):
                    wr = _regs(prv.split(None, 1)[1].split(",")[0]) if " " in prv else set()
                    if wr & (srcs | (_regs(ops[3]) if len(ops) > 3 else set())):
                        hits.append(f"{path}: `{prv}` {states} states before `{line}`")
                        break
                states += 1
            if states >= WAIT_S:
                break
    return hits


if __name__ == "__main__":
    found = [h for p in sys.argv[1:] for h in scan(p)]
    print("\n".join(found) if found else "no unpadded MFMA hazard")
    sys.exit(1 if found else 0)
