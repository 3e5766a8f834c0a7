#!/usr/bin/env python
"""Kernel timeline summary from a rocprofv3 --kernel-trace CSV: per kernel the mean duration and the mean idle gap before it
(end of the previous kernel on the device -> its start), over the last `--last` dispatches (steady state).
   python tools/timeline.py gpurun_out/prof/.../*_kernel_trace.csv [--last 600]"""
import argparse
import csv
import glob
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--last', type=int, default=600)
    args = ap.parse_args()
    path = sorted(glob.glob(args.csv))[-1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70]))
    rows.sort()
    rows = rows[-args.last:]
    stats = {}
    prev_end = None
    for s, e, name in rows:
        d = stats.setdefault(name, {'n': 0, 'dur_ns': 0.0, 'gap_ns': 0.0})
        d['n'] += 1
        d['dur_ns'] += e - s
        if prev_end is not None:
            d['gap_ns'] += max(0, s - prev_end)
        prev_end = max(prev_end or 0, e)
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    out = {'file': path, 'dispatches': len(rows), 'span_us': span/1e3, 'busy_frac': busy/span,
           'kernels': {k: {'n': v['n'], 'dur_us': v['dur_ns']/v['n']/1e3, 'gap_before_us': v['gap_ns']/v['n']/1e3}
                       for k, v in stats.items()}}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
