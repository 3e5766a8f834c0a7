#!/usr/bin/env python
"""Live VGPR count along the instruction stream of one kernel of a hipcc -S listing (straight-line approximation: branches are
ignored, which is right for kernels whose only branches skip cold code).  Shows where the register peak sits.
    python tools/vlive.py listing.s kernel_substring [--top N]"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]', tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 12
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if name in l and re.match(r'^[A-Za-z_][\w$]*:', l))
    body = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith('s_endpgm'):
            break
        if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
            continue
        body.append(t.split(';')[0].strip())
    ins = []
    for t in body:
        parts = t.split(None, 1)
        op = parts[0]
        ops = [x.strip() for x in parts[1].split(',')] if len(parts) > 1 else []
        if not ops:
            ins.append((t, [], []))
            continue
        store = op.startswith(('buffer_store', 'global_store', 'ds_write', 'flat_store', 'scratch_store', 'v_cmp', 's_'))
        if store:
            d, u = [], [r for o in ops for r in regs(o)]
        else:
            d, u = regs(ops[0]), [r for o in ops[1:] for r in regs(o)]
        ins.append((t, d, u))
    live = set()
    counts = [0]*len(ins)
    for i in range(len(ins) - 1, -1, -1):
        t, d, u = ins[i]
        for r in d:
            live.discard(r)
        for r in u:
            live.add(r)
        counts[i] = len(live)
    print('instructions', len(ins), 'peak live', max(counts))
    step = max(1, len(ins)//60)
    for i in range(0, len(ins), step):
        print('{:5d} {:4d}  {}'.format(i, max(counts[i:i + step]), ins[i][0][:90]))
    print('--- top')
    for i in sorted(range(len(ins)), key=lambda j: -counts[j])[:top]:
        print('{:5d} {:4d}  {}'.format(i, counts[i], ins[i][0][:100]))


if __name__ == '__main__':
    main()
