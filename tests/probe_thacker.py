import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/scratch')
import numpy as np
from thacker_cpu import setup
from oracle.ref_lib import RefSWE
n = int(sys.argv[1]); dt = float(sys.argv[2]); nmax = int(sys.argv[3])
mesh, bath, elev_v, av, lm = setup(n)
h = bath[mesh.cells]; al = av[mesh.cells]
ref = RefSWE(mesh.cell_xy(), mesh.cell_nbr, mesh.cell_nbr_facet, h, use_wetting_and_drying=True,
             wetting_and_drying_alpha=al, boundary_len=mesh.boundary_len)
eta = elev_v[mesh.cells].copy(); uv = np.zeros((mesh.num_cells, 3, 2))
p = mesh.cell_xy(); r = np.sqrt((p[:, :, 0] - lm/2)**2 + (p[:, :, 1] - lm/2)**2)
for k in range(nmax):
    uv, eta = ref.advance(uv, eta, dt, 1)
    sp = np.sqrt((uv**2).sum(axis=2))
    i = np.unravel_index(np.argmax(sp), sp.shape)
    H = h + eta; D = 0.5*(H + np.sqrt(H*H + al*al))
    if k % max(1, nmax//30) == 0 or sp.max() > 20:
        print('step %4d  max|u| %9.3f at r=%.0f km  H=%8.2f D=%7.3f alpha=%5.1f eta=%8.2f | minD %.4f' % (k, sp.max(), r[i]/1e3, H[i], D[i], al[i], eta[i], D.min()))
    if not np.isfinite(sp.max()) or sp.max() > 1e3: break
