#!/usr/bin/env python
"""profiles/rNN_traffic.json from the PMC passes of tools/pmc.sh on `tools/kbench.py --calibrate`: HBM bytes per stage-kernel
launch = FETCH_SIZE x correction + WRITE_SIZE, the correction taken from the calibration copy kernel of the same run (known
byte count; MI355X_MICROARCH.md: FETCH_SIZE counts half of the 8-B/lane fetches on gfx950).
   python tools/make_traffic_json.py <pmc outdir> <n_cells> <out.json> [note]"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    out, n_cells, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ''
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    mean = lambda name, c: sum(acc[name][c])/len(acc[name][c])
    cal = [k for k in acc if 'calibration_copy' in k]
    stride = (n_cells + 255)//256*256
    known = 9*stride*8
    res = {'source': 'tools/pmc.sh + tools/make_traffic_json.py (separate rocprofv3 --pmc passes, --kernel-trace only), '
                     'RectangleMesh with {:d} triangles, device ordering auto (Hilbert tiles). {:}'.format(n_cells, note)}
    fc, wc = 2.0, 1.0
    if cal:
        f_kb, w_kb = mean(cal[0], 'FETCH_SIZE'), mean(cal[0], 'WRITE_SIZE')
        fc, wc = known/(f_kb*1024.0), known/(w_kb*1024.0)
        res['calibration'] = {'kernel': 'swe_calibration_copy (8 B/lane coalesced)', 'known_bytes_read': known, 'FETCH_SIZE_KB': f_kb,
                              'WRITE_SIZE_KB': w_kb, 'fetch_correction': fc, 'write_correction_measured': wc, 'write_correction': 1.0}
    def issue(k):
        """wave-instructions of one launch (SQ_INSTS_VALU counts per wave, SQ_WAVES the waves), clock cycles of the launch"""
        d = {}
        for c in ('SQ_INSTS_VALU', 'SQ_WAVES', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SALU', 'GRBM_GUI_ACTIVE'):
            if acc[k].get(c):
                d[c] = mean(k, c)
        return d

    stage = sorted(k for k in acc if 'swe_stage_kernel<' in k)
    tot, launches = 0.0, 0
    for k in stage:
        f_kb, w_kb = mean(k, 'FETCH_SIZE'), mean(k, 'WRITE_SIZE')
        has_u0 = k.split('<')[1].split(',')[2].strip() == 'true'
        b = f_kb*1024.0*fc + w_kb*1024.0
        res['stage12_kernel' if has_u0 else 'stage0_kernel'] = dict({
            'name': k[:80], 'FETCH_SIZE_KB': f_kb, 'WRITE_SIZE_KB': w_kb, 'bytes': b,
            'algorithmic_bytes': (252.0 if has_u0 else 180.0)*n_cells, 'launches_sampled': len(acc[k]['FETCH_SIZE'])}, **issue(k))
        w = 2 if has_u0 else 1
        tot += w*b
        launches += w
    triple = sorted(k for k in acc if 'swe_fuse123_kernel<' in k)
    fused = sorted(k for k in acc if 'swe_fuse12_kernel<' in k)
    if triple:
        # all three stages in one launch (csrc/swe2d_fuse.h, swe_fuse123_kernel): a step is this launch; per element-update = / 3
        k = triple[0]
        f_kb, w_kb = mean(k, 'FETCH_SIZE'), mean(k, 'WRITE_SIZE')
        b = f_kb*1024.0*fc + w_kb*1024.0
        res['fused_stage_triple_kernel'] = dict({'name': k[:80], 'FETCH_SIZE_KB': f_kb, 'WRITE_SIZE_KB': w_kb, 'bytes': b,
                                                 'algorithmic_bytes_of_the_three_stage_launches': 684.0*n_cells,
                                                 'launches_sampled': len(acc[k]['FETCH_SIZE'])}, **issue(k))
        res.pop('stage0_kernel', None)
        res.pop('stage12_kernel', None)
        tot, launches = b, 3
        res['launches_per_step'] = 1
        if res['fused_stage_triple_kernel'].get('SQ_INSTS_VALU') is not None:
            res['valu_wave_instructions_per_step'] = res['fused_stage_triple_kernel']['SQ_INSTS_VALU']
    elif fused:
        # stages 1 + 2 in one launch (csrc/swe2d_fuse.h) + stage 3 as a stage launch: a step is these two; per element-update = / 3
        k = fused[0]
        f_kb, w_kb = mean(k, 'FETCH_SIZE'), mean(k, 'WRITE_SIZE')
        b = f_kb*1024.0*fc + w_kb*1024.0
        res['fused_stage_pair_kernel'] = dict({'name': k[:80], 'FETCH_SIZE_KB': f_kb, 'WRITE_SIZE_KB': w_kb, 'bytes': b,
                                               'algorithmic_bytes_of_the_two_stage_launches': (180.0 + 252.0)*n_cells,
                                               'launches_sampled': len(acc[k]['FETCH_SIZE'])}, **issue(k))
        s3 = res.get('stage12_kernel', {}).get('bytes', 0.0)
        res.pop('stage0_kernel', None)
        tot, launches = b + s3, 3
        res['launches_per_step'] = 2
        vi = [res[x].get('SQ_INSTS_VALU') for x in ('fused_stage_pair_kernel', 'stage12_kernel') if x in res]
        if len(vi) == 2 and all(v is not None for v in vi):
            res['valu_wave_instructions_per_step'] = vi[0] + vi[1]
    else:
        vi = [res[x].get('SQ_INSTS_VALU') for x in ('stage0_kernel', 'stage12_kernel') if x in res]
        if len(vi) == 2 and all(v is not None for v in vi):
            res['valu_wave_instructions_per_step'] = vi[0] + 2.0*vi[1]
    res['traffic_bytes_per_launch'] = tot/max(launches, 1)
    res['algorithmic_bytes_per_launch'] = 228.0*n_cells
    json.dump(res, open(dst, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
