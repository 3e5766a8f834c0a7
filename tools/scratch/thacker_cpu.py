"""Thacker paraboloid (test/swe2d/test_thacker.py) with the C restatement's explicit wetting-drying: stability scan."""
import sys, math, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from thetis_amd.mesh import RectangleMesh
from oracle.ref_lib import RefSWE

def setup(n):
    lm = 951646.46
    mesh = RectangleMesh(n, n, lm, lm)
    D0, L, eta0 = 50.0, 430620.0, 2.0
    A = ((D0 + eta0)**2 - D0**2)/((D0 + eta0)**2 + D0**2)
    X0 = Y0 = lm/2
    x, y = mesh.vertex_xy.T
    r2 = (x - X0)**2 + (y - Y0)**2
    bath = D0*(1 - r2/L**2)
    elev_v = D0*(math.sqrt(1 - A*A)/(1 - A) - 1 - r2*((1 + A)/(1 - A) - 1)/L**2)
    # automatic alpha as in solver2d
    p = mesh.cell_xy(); h = bath[mesh.cells]
    widths = np.abs(p - np.roll(p, 1, axis=1)).max(axis=1)
    d = p - p.mean(axis=1, keepdims=True)
    g = np.einsum('nij,nj->ni', np.linalg.pinv(d), h - h.mean(axis=1, keepdims=True))
    alpha_c = (widths*np.abs(g)).sum(axis=1)
    av = np.zeros(mesh.num_vertices)
    for i in range(3):
        np.maximum.at(av, mesh.cells[:, i], alpha_c)
    return mesh, bath, elev_v, av, lm

def l2err(mesh, eta, elev_v, lm):
    # masked L2 error as in the reference test (nodal P1 quadrature per cell)
    p = mesh.cell_xy(); X0 = lm/2
    r = np.sqrt((p[:, :, 0] - X0)**2 + (p[:, :, 1] - X0)**2)
    mask = 0.5*(1 - np.tanh((r - 420000.0)/1000.0))
    diff = mask*(eta - elev_v[mesh.cells])
    area = mesh.cell_areas()
    s = diff.sum(axis=1)
    integ = area/12.0*(s*s + (diff*diff).sum(axis=1))
    return math.sqrt(integ.sum())/lm

if __name__ == '__main__':
    n = int(sys.argv[1]); dt = float(sys.argv[2])
    mesh, bath, elev_v, av, lm = setup(n)
    print('n', n, 'cells', mesh.num_cells, 'alpha range', av.min(), av.max(), 'dx', lm/n)
    ref = RefSWE(mesh.cell_xy(), mesh.cell_nbr, mesh.cell_nbr_facet, bath[mesh.cells], use_wetting_and_drying=True,
                 wetting_and_drying_alpha=av[mesh.cells], boundary_len=mesh.boundary_len)
    import ctypes
    if len(sys.argv) > 3:
        ref.lib.swe2d_ref_set_wd_tau.argtypes = [ctypes.c_double]
        ref.lib.swe2d_ref_set_wd_tau(float(sys.argv[3]))
    eta = elev_v[mesh.cells].copy(); uv = np.zeros((mesh.num_cells, 3, 2))
    T = 43200.0
    nsteps = int(round(T/dt)); chunk = max(1, nsteps//24)
    t = 0
    for k in range(0, nsteps, chunk):
        m = min(chunk, nsteps - k)
        uv, eta = ref.advance(uv, eta, dt, m)
        t += m*dt
        if not np.isfinite(eta).all():
            print('blew up before t =', t); break
        print('t %7.0f  max|u| %8.3f  eta [%8.3f, %8.3f]  err %.4f' % (t, np.abs(uv).max(), eta.min(), eta.max(), l2err(mesh, eta, elev_v, lm)))
