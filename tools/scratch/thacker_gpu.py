import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from test_wetting_drying import thacker_case, thacker_error, _auto_alpha
from thetis_amd.device import Swe2dDevice
from oracle.ref_lib import RefSWE
from oracle.swe2d_oracle import SWEOracle
n, dt = 25, 50.0
mesh, bath, elev_v, lm = thacker_case(n)
av = np.minimum(_auto_alpha(mesh, bath), 2.0)
h, al = bath[mesh.cells], av[mesh.cells]
dev = Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len)
dev.set_wetting_and_drying(av)
uv0, eta0 = np.zeros((mesh.num_cells, 3, 2)), elev_v[mesh.cells].copy()
dev.set_state(uv0, eta0)
print('dev vol after set_state', dev.diagnostics()[2:])
orc = SWEOracle(mesh.vertex_xy, mesh.cells, bath, use_wetting_and_drying=True, wd_mode='nodal', wetting_and_drying_alpha=av)
ec = orc.wd_clip_state(eta0)
print('clip diff dev-oracle', np.abs(dev.get_state()[1] - ec).max(), 'oracle vol', orc.wd_volume(ec), orc.wd_volume(eta0))
ref = RefSWE(mesh.cell_xy(), mesh.cell_nbr, mesh.cell_nbr_facet, h, use_wetting_and_drying=True, wetting_and_drying_alpha=al, boundary_len=mesh.boundary_len)
u, e = uv0, ec
for k in range(5):
    dev.advance(1); u, e = ref.advance(u, e, dt, 1)
    ud, ed = dev.get_state()
    print(k, 'dev vol', dev.diagnostics()[2], 'ref vol', orc.wd_volume(e), 'diff eta', np.abs(ed - e).max(), 'diff u', np.abs(ud - u).max())
