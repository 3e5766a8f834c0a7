#!/usr/bin/env python
"""Register / scratch / LDS use of the kernels of one translation unit, by the compiler's own remarks:
    python tools/kres.py swe2d_k_flow.hip [-DSWE_...] [--filter flow_kernel]
(hipcc -Rpass-analysis=kernel-resource-usage on thetis_amd/csrc/<unit>; one line per kernel)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    filt = None
    if '--filter' in args:
        i = args.index('--filter')
        filt = args[i + 1]
        del args[i:i + 2]
    unit = args[0]
    extra = args[1:]
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-c',
           os.path.join(ROOT, 'thetis_amd', 'csrc', unit), '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for line in err.splitlines():
        m = re.search(r'remark: (?:[^:]*: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        else:
            cur[k.split(' ')[0]] = v
    try:
        names = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
    except OSError:
        names = [r['name'] for r in rows]
    for r, n in zip(rows, names):
        n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('(anonymous namespace)::', '')
        if filt and filt not in n:
            continue
        print('{:60s} VGPR {:>3s} AGPR {:>3s} scratch {:>4s} occ {:>2s} LDS {:>6s}'.format(n[:60], r.get('VGPRs', '?'), r.get('AGPRs', '?'),
              r.get('ScratchSize', '?'), r.get('Occupancy', '?'), r.get('LDS', '?')))


if __name__ == '__main__':
    main()
