import sys, math, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from thetis_amd.mesh import RectangleMesh
from oracle.ref_lib import RefSWE
LX, LY = 13800.0, 7200.0
nx, ny, dt = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
floor = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
mesh = RectangleMesh(nx, ny, LX, LY)
bath = mesh.vertex_xy[:, 0]/2760.0
h = bath[mesh.cells]
uv = np.zeros((mesh.num_cells, 3, 2)); eta = np.zeros((mesh.num_cells, 3))
D = lambda e: 0.5*((h + e) + np.sqrt((h + e)**2 + 0.4**2))
area = mesh.cell_areas()
t = 0.0; chunk = max(1, int(600/dt))
first = True
for k in range(0, int(43200/dt), chunk):
    elev = -2.0*math.sin(2*math.pi*(t + 0.5*chunk*dt)/43200.0)
    ref = RefSWE(mesh.cell_xy(), mesh.cell_nbr, mesh.cell_nbr_facet, h, manning_drag_coefficient=0.02,
                 use_wetting_and_drying=True, wetting_and_drying_alpha=0.4, bnd_conditions={2: {'elev': elev}},
                 boundary_len=mesh.boundary_len)
    uv, eta = ref.advance(uv, eta, dt, chunk)
    t += chunk*dt
    if not np.isfinite(eta).all():
        print('blew up before', t); break
    if (k//chunk) % 12 == 11:
        print('t %6.0f tide %6.2f max|u| %6.3f eta [%7.3f %7.3f] minH %7.3f minD %.4f wet frac %.3f' % (t, elev, np.abs(uv).max(), eta.min(), eta.max(), (h+eta).min(), D(eta).min(), ((h+eta).mean(axis=1) > 0.05).mean()))
