import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from helpers import channel_case
from thetis_amd.device import Swe2dDevice
mesh, bath, uv, eta = channel_case(nx=24, ny=9, seed=17, amp_eta=0.3, amp_u=0.2)
out = {}
for force in ('0', '1'):
    os.environ['THETIS_AMD_BND_INLINE'] = force
    dev = Swe2dDevice(mesh, bath, 2.0, boundary_len=mesh.boundary_len)
    dev.set_state(uv, eta)
    dev.solve_stage(0)
    out[force] = dev.get_state(0)
    dev.close()
du = np.abs(out['0'][0] - out['1'][0]).max(axis=(1, 2))
de = np.abs(out['0'][1] - out['1'][1]).max(axis=1)
bnd = (mesh.cell_nbr < 0).any(axis=1)
print('cells', mesh.num_cells, 'boundary cells', bnd.sum())
print('differing cells (u):', (du > 0).sum(), 'of which boundary', (du[bnd] > 0).sum(), 'max', du.max(), 'rel', du.max()/np.abs(out['0'][0]).max())
print('differing cells (eta):', (de > 0).sum(), 'of which boundary', (de[bnd] > 0).sum(), 'max', de.max())
nbf = (mesh.cell_nbr < 0).sum(axis=1)
for nb in (1, 2):
    sel = nbf == nb
    print('cells with', nb, 'boundary facets:', sel.sum(), 'differing u', (du[sel] > 0).sum(), 'eta', (de[sel] > 0).sum())
