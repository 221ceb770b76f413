"""Scan hipcc's gfx950 assembly for a >64-bit VMEM store whose data VGPRs are written by a VALU within two wait states
(fall-through path and branch targets).  usage: python scan_store_hazard.py file.s ..."""
import re, sys
def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
def scan(path):
    """-> list of 'file:line: store -> overwriting instruction' strings"""
    L = [l.rstrip() for l in open(path)]
    labels = {l.split(':')[0]: i for i, l in enumerate(L) if re.match(r'^\.LBB\d+_\d+:', l)}

    def real(i):
        out = []
        while i < len(L) and len(out) < 4:
            t = L[i].strip()
            if t and not t.startswith(';') and not t.startswith('.') and not re.match(r'^\.?\w+:', t):
                out.append((i, t))
            i += 1
        return out
    hits = []
    for i, l in enumerate(L):
        t = l.strip()
        m = re.match(r'(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)', t)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(',')]
        data = regs(ops[0]) if m.group(1).startswith('buffer') else regs(ops[1])
        paths = [real(i + 1)]
        for (_, nt) in paths[0][:2]:
            b = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', nt)
            if b and b.group(1) in labels:
                paths.append(real(labels[b.group(1)]))
        for p in paths:
            states = 0
            for (j, nt) in p:
                if states >= 2:
                    break
                op = nt.split()[0]
                if op == 's_nop':
                    states += int(nt.split()[1]) + 1
                    continue
                if op.startswith('v_') and not op.startswith('v_cmp') and not op.startswith('v_mfma'):
                    dst = regs(nt.split()[1].rstrip(','))
                    if dst & data:
                        hits.append(f"{path}:{i + 1}: {t} -> line {j + 1}: {nt}")
                        break
                states += 1
    return hits


if __name__ == "__main__":
    for path in sys.argv[1:]:
        h = scan(path)
        print("\n".join(h))
        print(path, "hits", len(h))
