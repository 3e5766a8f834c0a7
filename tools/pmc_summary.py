#!/usr/bin/env python
"""Summarise tools/pmc.sh output: per-kernel mean of every counter.  Usage: tools/pmc_summary.py <outdir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else 'swe_stage_kernel'
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        if sub not in name:
            continue
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for name, cs in sorted(acc.items()):
    print(name)
    for c, v in sorted(cs.items()):
        print('   {:32s} mean {:16.1f}   n={:d}'.format(c, sum(v)/len(v), len(v)))
