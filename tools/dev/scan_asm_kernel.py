#!/usr/bin/env python3
"""Scan hand-written / generated gfx950 assembly (csrc/asm/gen_wino4_asm.py's output) for the wait states the ASSEMBLER never pads
(it is not a compiler: every hazard of LLVM's GCNHazardRecognizer for gfx940+ is the author's job).  Checked, along every path of the
control-flow graph (branch targets and back-edges included), counting one state per instruction and N + 1 per `s_nop N`:

  * a VALU write of a VGPR needs 1 state before v_readfirstlane / v_readlane reads it        (found the hard way: the wave number of
    conv3x3_wino4a_f32 was whatever the register held before -- tools/dev/asm_probes/probe_ids.s)
  * a VALU write of an SGPR / VCC needs 2 states before another VALU reads it                 (v_cmp -> v_cndmask, v_readfirstlane -> VALU)
  * a VALU write of an SGPR needs 5 states before a VMEM instruction reads it                 (descriptor, soffset)
  * an SALU write of M0 needs 1 state before an LDS-DMA load (`... lds`) uses it
  * an MFMA result needs 11 states (8-pass v_mfma_f32_16x16x4_f32) before anything but the accumulate chain touches it
usage: scan_asm_kernel.py file.s"""
import re
import sys

_REG = re.compile(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b|\b(vcc|m0|exec)\b")


def regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4):
            out.add((m.group(4), int(m.group(5))))
        else:
            out.add((m.group(6), 0))
    return out


def parse(path):
    ins, labels = [], {}
    for raw in open(path):
        line = raw.split("//")[0].split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(ins)
            continue
        if line.startswith(".") or line.startswith("---") or ":" in line.split()[0]:
            continue
        if not re.match(r"^[svdgb][a-z_]", line):
            continue
        ins.append(line)
    return ins, labels


def succ(ins, labels, j):
    line = ins[j]
    op = line.split()[0]
    if op == "s_endpgm":
        return []
    if op == "s_branch" or op.startswith("s_cbranch"):
        t = labels.get(line.split()[-1])
        out = [] if t is None else [t]
        if op != "s_branch" and j + 1 < len(ins):
            out.append(j + 1)
        return out
    return [j + 1] if j + 1 < len(ins) else []


def states(line):
    m = re.match(r"s_nop\s+(\d+)", line)
    return int(m.group(1)) + 1 if m else 1


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_mfma")


def is_vmem(op):
    return op.startswith(("buffer_", "global_", "flat_", "scratch_"))


def dst_and_srcs(line):
    op, _, rest = line.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    return op, regs(ops[0]) if ops else set(), regs(",".join(ops[1:])) if len(ops) > 1 else set()


def scan(path):
    ins, labels = parse(path)
    hits = []

    def forward(i, need, bad, what):
        """every path out of instruction i for `need` states: bad(line) -> True is a hazard"""
        best = {}
        work = [(k, 0) for k in succ(ins, labels, i)]
        while work:
            j, st = work.pop()
            if st >= need or best.get(j, need) <= st:
                continue
            best[j] = st
            if bad(ins[j]):
                hits.append(f"{what}: `{ins[i]}` then after {st} state(s) `{ins[j]}`")
                return
            for k in succ(ins, labels, j):
                work.append((k, st + states(ins[j])))

    for i, line in enumerate(ins):
        op, dst, srcs = dst_and_srcs(line)
        if is_valu(op):
            vdst = {r for r in dst if r[0] == "v"}
            sdst = {r for r in dst if r[0] in ("s", "vcc")}
            if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")) or sdst:
                if op.startswith("v_cmp") and not sdst:
                    sdst = {("vcc", 0)}
                forward(i, 2, lambda n: is_valu(n.split()[0]) and bool((dst_and_srcs(n)[2] | (({("vcc", 0)}) if n.split()[0].startswith(("v_cndmask_b32_e32", "v_cndmask_b32 ")) and "vcc" in n else set())) & sdst), "VALU-written SGPR read by a VALU")
                forward(i, 5, lambda n: is_vmem(n.split()[0]) and bool(regs(n) & sdst), "VALU-written SGPR read by VMEM")
            if vdst:
                forward(i, 1, lambda n: n.split()[0].startswith(("v_readfirstlane", "v_readlane")) and bool(dst_and_srcs(n)[2] & vdst), "VALU-written VGPR read by v_readfirstlane")
        elif op.startswith("s_") and ("m0", 0) in dst:
            forward(i, 1, lambda n: n.rstrip().endswith(" lds") or " lds " in n, "M0 written right in front of an LDS-DMA load")
        elif op.startswith("v_mfma"):
            ops = [o.strip() for o in line.split(None, 1)[1].split(",")]
            d = regs(ops[0])

            def touches(n, d=d):
                nop = n.split()[0]
                if nop.startswith("v_mfma"):
                    o = [x.strip() for x in n.split(None, 1)[1].split(",")]
                    return bool((regs(o[1]) | regs(o[2])) & d) or (bool(regs(o[0]) & d) and regs(o[3]) != regs(o[0]))
                return (not nop.startswith("s_")) and bool(regs(n) & d)
            forward(i, 11, touches, "MFMA result touched too early")
    return hits


if __name__ == "__main__":
    found = [h for p in sys.argv[1:] for h in scan(p)]
    print("\n".join(found[:40]) if found else "no unpadded hazard")
    sys.exit(1 if found else 0)
