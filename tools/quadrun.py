#!/usr/bin/env python
"""A few SSPRK33 steps on an nx x nx quadrilateral mesh (profiling target).  python tools/quadrun.py <nx> <steps>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thetis_amd.device import Swe2dDevice        # noqa: E402
from thetis_amd.mesh import RectangleMesh        # noqa: E402

nx, steps = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1234)
m = RectangleMesh(nx, nx, 100e3, 100e3, quadrilateral=True)
c = m.cell_xy()
eta = 0.5*np.exp(-((c[:, :, 0] - 50e3)**2 + (c[:, :, 1] - 50e3)**2)/(5e3)**2)
dev = Swe2dDevice(m, np.full(m.num_vertices, 20.0), 0.25)
dev.set_state(1e-3*rng.uniform(-1, 1, size=(m.num_cells, 4, 2)), eta)
dev.advance(steps)
dev.synchronize()
print('done', m.num_cells)
