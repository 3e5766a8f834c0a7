import sys, math, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/scratch')
import numpy as np
from thacker_cpu import setup, l2err
from oracle.ref_lib import RefSWE
n = int(sys.argv[1]); dt = float(sys.argv[2]); tau = float(sys.argv[3])
mesh, bath, elev_v, av, lm = setup(n)
if len(sys.argv) > 4:
    av = np.minimum(av, float(sys.argv[4]))
h = bath[mesh.cells]; al = av[mesh.cells]
ref = RefSWE(mesh.cell_xy(), mesh.cell_nbr, mesh.cell_nbr_facet, h, use_wetting_and_drying=True,
             wetting_and_drying_alpha=al, boundary_len=mesh.boundary_len)
import os
eta = elev_v[mesh.cells].copy(); uv = np.zeros((mesh.num_cells, 3, 2))
p = mesh.cell_xy(); r = np.sqrt((p[:, :, 0] - lm/2)**2 + (p[:, :, 1] - lm/2)**2)
ic = np.unravel_index(np.argmin(r), r.shape)
D0, L, eta0 = 50.0, 430620.0, 2.0
A = ((D0 + eta0)**2 - D0**2)/((D0 + eta0)**2 + D0**2)
om = math.sqrt(8*9.81*D0)/L
nsteps = int(round(43200/dt)); chunk = nsteps//12
t = 0.0
vol0 = None
for k in range(12):
    uv, eta = ref.advance(uv, eta, dt, chunk); t += chunk*dt
    r0 = r[ic]
    ana = D0*(math.sqrt(1 - A*A)/(1 - A*math.cos(om*t)) - 1 - r0**2/L**2*((1 - A*A)/(1 - A*math.cos(om*t))**2 - 1))
    H = h + eta; D = 0.5*(H + np.sqrt(H*H + al*al))
    area = mesh.cell_areas(); vol = (area*D.mean(axis=1)).sum()
    vol0 = vol0 or vol
    print('t %6.0f center eta %7.3f analytic %7.3f  vol drift %.2e  err %.4f' % (t, eta[ic], ana, vol/vol0 - 1, l2err(mesh, eta, elev_v, lm)))
