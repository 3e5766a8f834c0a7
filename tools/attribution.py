#!/usr/bin/env python
"""
Static instruction attribution of one gfx950 kernel with inline stacks: which source function, and which line of the kernel body,
owns how many VALU / SALU / LDS / memory instructions.

    cd /tmp
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -gline-tables-only $REPO/thetis_amd/csrc/swe2d_unity.hip -o swe2d_dev.o
    clang-offload-bundler --unbundle --type=o --input=swe2d_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=swe2d_gfx950.co
    python tools/attribution.py swe2d_gfx950.co <mangled kernel symbol> [--by callee|kline|innermost] [--regions file]

(-gline-tables-only keeps the inlined-subroutine records and does not change the generated code.)  Every instruction is looked up
with llvm-symbolizer --inlines; it is charged
  --by callee     to the function the kernel body calls (the outermost inlined frame below the kernel; 'kernel body' if none),
  --by kline      to the line of the kernel body the (possibly inlined) code was called from,
  --by innermost  to the innermost frame that is not a header intrinsic wrapper (fma, fabs ... of __clang_hip_math.h).
``--regions``: a text file of "first_line last_line label" rows; with ``--by kline`` the lines are summed per region.
Counts are static, per wave; the stage kernels are straight-line code up to the boundary branches, so static ~ executed for
interior cells and an upper bound for the rest.
"""
import collections
import re
import subprocess
import sys

LLVM = '/opt/rocm/lib/llvm/bin/'


def classify(op):
    if op.startswith('v_'):
        if re.match(r'v_(rcp|rsq|sqrt|log|exp|sin|cos)_', op):
            return 'trans'
        return 'valu_f64' if 'f64' in op else 'valu'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('scratch_'):
        return 'scratch'
    if op.startswith(('buffer_', 'global_', 'flat_')):
        return 'vmem'
    return 'other'


def main():
    obj, symbol = sys.argv[1], sys.argv[2]
    by = sys.argv[sys.argv.index('--by') + 1] if '--by' in sys.argv else 'callee'
    regions = []
    if '--regions' in sys.argv:
        for l in open(sys.argv[sys.argv.index('--regions') + 1]):
            p = l.split(None, 2)
            if len(p) == 3 and p[0].isdigit():
                regions.append((int(p[0]), int(p[1]), p[2].strip()))
    dis = subprocess.run([LLVM + 'llvm-objdump', '-d', '--disassemble-symbols=' + symbol, obj], capture_output=True, text=True).stdout
    insts = []
    for l in dis.splitlines():
        m = re.match(r'\s+(\S+)\s.*//\s*([0-9A-F]{12}):', l)
        if m:
            insts.append((int(m.group(2), 16), m.group(1)))
    if not insts:
        sys.exit('symbol not found in ' + obj)
    q = '\n'.join('0x{:x}'.format(a) for a, _ in insts) + '\n'
    sym = subprocess.run([LLVM + 'llvm-symbolizer', '--obj=' + obj, '--inlines', '--functions=short'], input=q, capture_output=True, text=True).stdout
    stacks = []
    for block in sym.strip().split('\n\n'):
        ls = block.strip().splitlines()
        stacks.append([(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)])
    assert len(stacks) == len(insts), (len(stacks), len(insts))
    agg = collections.defaultdict(collections.Counter)
    for (addr, op), st in zip(insts, stacks):
        kind = classify(op)
        kframe = st[-1]                                   # the kernel itself
        kline = int(kframe[1].rsplit(':', 2)[-2]) if kframe[1].count(':') >= 2 else 0
        if by == 'kline':
            key = kline
            if regions:
                key = next((lab for a, b, lab in regions if a <= kline <= b), 'line {:d} (no region)'.format(kline))
        elif by == 'innermost':
            fr = next((f for f in st if '__clang_hip_math' not in f[1] and 'amd_detail' not in f[1]), st[-1])
            key = fr[0] if fr is not kframe else 'kernel body'
        else:
            key = st[-2][0] if len(st) >= 2 else 'kernel body'
            if len(st) >= 2 and ('__clang_hip_math' in st[-2][1] or 'amd_detail' in st[-2][1]):
                key = 'kernel body'                       # fma(), fabs() ... called from the kernel body itself
        agg[key][kind] += 1
    kinds = ['valu_f64', 'trans', 'valu', 'salu', 'lds', 'vmem', 'scratch', 'other']
    total = collections.Counter()
    for c in agg.values():
        total.update(c)
    nv = lambda c: c['valu'] + c['valu_f64'] + c['trans']
    print('{:<58s}'.format(symbol[:58]) + ''.join('{:>9s}'.format(k) for k in kinds) + '{:>9s}{:>7s}'.format('all_valu', '%'))
    rows = sorted(agg.items(), key=(lambda kv: kv[0]) if by == 'kline' and not regions else (lambda kv: -nv(kv[1])))
    for key, c in rows:
        print('{:<58s}'.format(str(key)[:58]) + ''.join('{:>9d}'.format(c[k]) for k in kinds)
              + '{:>9d}{:>7.1f}'.format(nv(c), 100.0*nv(c)/max(1, nv(total))))
    print('{:<58s}'.format('TOTAL') + ''.join('{:>9d}'.format(total[k]) for k in kinds) + '{:>9d}{:>7.1f}'.format(nv(total), 100.0))


if __name__ == '__main__':
    main()
