#!/usr/bin/env python
"""Writes tests/golden/firedrake_like_quad_mesh.json: 2 x 2 unit squares in the array shapes a Firedrake quadrilateral mesh hands
out (FIAT tensor-product cells, [FD-assumed] - Firedrake cannot be imported here).  The expected
counter-clockwise cells and DQ-1 permutation are derived from the geometry (angle sort around the cell centre), not with the
adapter under test (thetis_amd/firedrake_adapter.py)."""
import json
import os

import numpy as np

coords = [[i, j] for j in range(3) for i in range(3)]


def vid(i, j):
    return 3*j + i


cells = []
for cj in range(2):
    for ci in range(2):
        # lexicographic reference order (0,0), (0,1), (1,0), (1,1): first reference direction along x, second along y
        cells.append([vid(ci, cj), vid(ci, cj + 1), vid(ci + 1, cj), vid(ci + 1, cj + 1)])
cells[2] = [vid(0, 1), vid(1, 1), vid(0, 2), vid(1, 2)]          # mirrored cell: first reference direction along y
FACET_VERTICES = [[0, 1], [2, 3], [0, 2], [1, 3]]                 # FIAT tensor-product quadrilateral
edge = {}
for c, cv in enumerate(cells):
    for f, (a, b) in enumerate(FACET_VERTICES):
        edge.setdefault(tuple(sorted((cv[a], cv[b]))), []).append((c, f))
int_fc, int_lf, ext_fc, ext_lf, ext_mk = [], [], [], [], []
for e, lst in sorted(edge.items()):
    if len(lst) == 2:
        int_fc.append([lst[0][0], lst[1][0]])
        int_lf.append([lst[0][1], lst[1][1]])
    else:
        (c, f), = lst
        (x0, y0), (x1, y1) = coords[e[0]], coords[e[1]]
        ext_fc.append(c)
        ext_lf.append(f)
        ext_mk.append(1 if x0 == x1 == 0 else 2 if x0 == x1 == 2 else 3 if y0 == y1 == 0 else 4)
dg = [[4*c + i for i in range(4)] for c in range(4)]
dg[1], dg[2] = [7, 4, 6, 5], [9, 8, 11, 10]                      # DQ-1 dofs scrambled inside cells 1 and 2


def ccw(cv):
    pts = np.array([coords[v] for v in cv], float)
    ctr = pts.mean(0)
    order = list(np.argsort(np.arctan2(pts[:, 1] - ctr[1], pts[:, 0] - ctr[0])))
    k = order.index(0)
    return order[k:] + order[:k]                                  # counter-clockwise, starting at the first listed vertex


orders = [ccw(cv) for cv in cells]
fx = {'_comment': 'Hand-written (make_firedrake_like_quad_mesh.py): 2 x 2 unit squares as FIAT tensor-product quadrilaterals: vertices '
                  '3*j + i at (i, j); local vertices in lexicographic reference order (0,0),(0,1),(1,0),(1,1), first reference direction '
                  'along x for cells 0, 1, 3 and along y (mirrored cell) for cell 2; local facets 0:{0,1} 1:{2,3} 2:{0,2} 3:{1,3}; '
                  'markers 1: x=0, 2: x=2, 3: y=0, 4: y=2; DQ-1 dofs cell by cell, scrambled inside cells 1 and 2.',
      'coords': coords, 'cell_vertices': cells, 'int_facet_cell': int_fc, 'int_local_facet': int_lf, 'ext_facet_cell': ext_fc,
      'ext_local_facet': ext_lf, 'ext_markers': ext_mk, 'dg_cell_nodes': dg,
      'expected_cell_vertices_ccw': [[cv[i] for i in o] for cv, o in zip(cells, orders)],
      'expected_dg_perm': [[d[i] for i in o] for d, o in zip(dg, orders)]}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'firedrake_like_quad_mesh.json'), 'w') as f:
    json.dump(fx, f, indent=1)
