"""
Equatorial Rossby soliton set-up of test/swe2d/test_rossby_wave.py (Huang et al. 2008, JGR 113(C7), p.3-6):
first-order asymptotic initial condition (:22-137) and the error metrics of ``run`` (:200-222).
Test data only - shared by the CPU (oracle) and GPU tests.
"""
import numpy as np

_U = {0: 1.7892760e+00, 2: 0.1164146e+00, 4: -0.3266961e-03, 6: -0.1274022e-02, 8: 0.4762876e-04, 10: -0.1120652e-05,
      12: 0.1996333e-07, 14: -0.2891698e-09, 16: 0.3543594e-11, 18: -0.3770130e-13, 20: 0.3547600e-15,
      22: -0.2994113e-17, 24: 0.2291658e-19, 26: -0.1178252e-21}
_V = {3: -0.6697824e-01, 5: -0.2266569e-02, 7: 0.9228703e-04, 9: -0.1954691e-05, 11: 0.2925271e-07, 13: -0.3332983e-09,
      15: 0.2916586e-11, 17: -0.1824357e-13, 19: 0.4920951e-16, 21: 0.6302640e-18, 23: -0.1289167e-19, 25: 0.1471189e-21}
_E = {0: -3.0714300e+00, 2: -0.3508384e-01, 4: -0.1861060e-01, 6: -0.2496364e-03, 8: 0.1639537e-04, 10: -0.4410177e-06,
      12: 0.8354759e-09, 14: -0.1254222e-09, 16: 0.1573519e-11, 18: -0.1702300e-13, 20: 0.1621976e-15,
      22: -0.1382304e-17, 24: 0.1066277e-19, 26: -0.1178252e-21}


def _hermite_sum(y, coeffs):
    polys = [np.ones_like(y), 2*y]
    for i in range(2, 28):
        polys.append(2*y*polys[i - 1] - 2*(i - 1)*polys[i - 2])
    return sum(c*polys[i] for i, c in coeffs.items())


def asymptotic_uv(x, y, time=0.0, B=0.395):
    c = -1.0/3.0 - 0.395*B*B
    xi = x - c*time
    psi = np.exp(-0.5*y*y)
    phi = 0.771*(B/np.cosh(B*xi))**2
    dphidx = -2*B*phi*np.tanh(B*xi)
    C = -0.395*B*B
    u = phi*0.25*(-9 + 6*y*y)*psi + C*phi*0.5625*(3 + 2*y*y)*psi + phi*phi*psi*_hermite_sum(y, _U)
    v = 2*y*dphidx*psi + dphidx*phi*psi*_hermite_sum(y, _V)
    return u, v


def asymptotic_elev(x, y, time=0.0, B=0.395):
    c = -1.0/3.0 - 0.395*B*B
    xi = x - c*time
    psi = np.exp(-0.5*y*y)
    phi = 0.771*(B/np.cosh(B*xi))**2
    C = -0.395*B*B
    return phi*0.25*(3 + 6*y*y)*psi + C*phi*0.5625*(-5 + 2*y*y)*psi + phi*phi*psi*_hermite_sum(y, _E)


def rossby_mesh(level):
    from thetis_amd.mesh import PeriodicRectangleMesh
    lx, ly = 48.0, 24.0
    mesh = PeriodicRectangleMesh(2*level, level, lx, ly, direction='x')
    mesh.vertex_xy = np.ascontiguousarray(mesh.vertex_xy - np.array([lx/2, ly/2]))     # test_rossby_wave.py:150-151
    return mesh


def metrics(cell_xy, eta):
    """h+, h-, c+, c- relative to the high-resolution FVCOM values 0.1567020 and 47.18 (test_rossby_wave.py:203-222)."""
    x, y = cell_xy[:, :, 0].ravel(), cell_xy[:, :, 1].ravel()
    e = (np.sign(y)*eta.ravel())
    i_n, i_s = np.argmax(e), np.argmin(e)
    h_n, h_s = e[i_n]/0.1567020, e[i_s]/-0.1567020
    c_n, c_s = (48.0 - x[i_n])/47.18, (48.0 - x[i_s])/47.18
    return h_n, h_s, c_n, c_s


def check_convergence(m24, m48, rtol=0.02):
    for a, b in zip(m24, m48):
        slope = (1 - abs(1 - b))/(1 - abs(1 - a))
        assert slope > 1.0 - rtol, (m24, m48)
