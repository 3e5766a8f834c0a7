#!/usr/bin/env python
"""Generates tests/golden/coast.msh: a small unstructured coastal mesh in Gmsh MSH 2.2 ASCII (the format of the reference's
demos/north_sea.msh, which cannot be committed): graded Delaunay triangulation of a bay with a curved coastline and an island,
physical ids 100 (open sea boundary x = 0), 200 (coast), 300 (island), as a mesh generator would number it (no locality in the
cell / vertex numbering).   python tests/golden/make_coast_mesh.py"""
import os
import sys

import numpy as np
from scipy.spatial import Delaunay

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thetis_amd.mesh import Mesh2d          # noqa: E402
from thetis_amd.meshio import write_gmsh     # noqa: E402

LX, LY = 60e3, 40e3
coast = lambda x: 30e3 + 4e3*np.sin(x/8e3)                                 # land above this line
island = lambda x, y: ((x - 35e3)/7e3)**2 + ((y - 14e3)/4e3)**2            # < 1 inside the island


def main():
    rng = np.random.default_rng(2026)
    # graded point cloud: rejection sampling with a density that grows towards the coast and the island
    pts = []
    while len(pts) < 1500:
        x, y = rng.uniform(0, LX), rng.uniform(0, LY)
        if y > coast(x) - 300.0 or island(x, y) < 1.08:
            continue
        d = min(coast(x) - y, 7e3*(np.sqrt(island(x, y)) - 1.0))
        if rng.uniform() < 0.15 + 0.85*np.exp(-d/6e3):
            pts.append((x, y))
    # boundary points: open boundary, bottom, right side, coastline, island
    s = np.linspace(0, 1, 41)
    pts += [(0.0, v) for v in coast(0.0)*s] + [(LX, v) for v in coast(LX)*s[1:]] + [(v, 0.0) for v in LX*s[1:-1]]
    xc = np.linspace(0, LX, 90)[1:-1]
    pts += list(zip(xc, coast(xc)))
    th = np.linspace(0, 2*np.pi, 56, endpoint=False)
    pts += list(zip(35e3 + 7e3*np.cos(th), 14e3 + 4e3*np.sin(th)))
    p = np.array(pts)
    tri = Delaunay(p).simplices
    c = p[tri].mean(axis=1)
    keep = (c[:, 1] < coast(c[:, 0])) & (island(c[:, 0], c[:, 1]) > 1.0)
    tri = tri[keep]
    a, b, cc = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    area2 = (b[:, 0] - a[:, 0])*(cc[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1])*(cc[:, 0] - a[:, 0])
    # drop slivers on the boundary curves: smallest altitude (2A / longest edge) at least 120 m
    e = np.stack([np.hypot(*(b - a).T), np.hypot(*(cc - b).T), np.hypot(*(a - cc).T)], axis=1)
    tri = tri[np.abs(area2)/e.max(axis=1) > 120.0]
    tri = tri[rng.permutation(len(tri))]                                      # a mesh generator's numbering: no locality
    used = np.unique(tri)
    remap = np.full(len(p), -1)
    remap[used] = rng.permutation(len(used))
    xy = np.empty((len(used), 2))
    xy[remap[used]] = p[used]
    mesh = Mesh2d(xy, remap[tri], name='coast')
    k = 3
    ci, fi = np.nonzero(mesh.cell_nbr < 0)
    for c_, f_ in zip(ci, fi):
        m = 0.5*(mesh.vertex_xy[mesh.cells[c_, f_]] + mesh.vertex_xy[mesh.cells[c_, (f_ + 1) % k]])
        if m[0] < 1.0:
            tag = 100
        elif island(m[0], m[1]) < 1.3:
            tag = 300
        else:
            tag = 200
        mesh.cell_nbr[c_, f_] = -tag
    mesh.boundary_len = mesh._boundary_length()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'coast.msh')
    write_gmsh(mesh, out)
    p3 = mesh.cell_xy()
    ed = np.stack([np.hypot(*(p3[:, (i + 1) % 3] - p3[:, i]).T) for i in range(3)], axis=1)
    print(out, mesh.num_cells, 'triangles', mesh.num_vertices, 'vertices', mesh.boundary_len, 'min altitude',
          (2*mesh.cell_areas()/ed.max(axis=1)).min())


if __name__ == '__main__':
    main()
