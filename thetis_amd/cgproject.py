"""
CG-P1 L2 projections used only to pick the automatic CFL time step at setup
(thetis/solver2d.py:149-177, thetis/utility.py:620-640): ``solve(inner(test, trial)*dx == inner(test, f)*dx)`` on the
continuous P1 space.  Host-side (scipy sparse), runs once.
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.linalg import spsolve

from .function import cell_quadrature


def _p1_mass(mesh):
    cells = mesh.cells
    area = mesh.cell_areas()
    k = cells.shape[1]
    if k == 3:
        m_loc = np.array([[2.0, 1.0, 1.0], [1.0, 2.0, 1.0], [1.0, 1.0, 2.0]])/12.0
    else:
        m_loc = np.array([[4.0, 2.0, 1.0, 2.0], [2.0, 4.0, 2.0, 1.0], [1.0, 2.0, 4.0, 2.0], [2.0, 1.0, 2.0, 4.0]])/36.0
    rows = np.repeat(cells, k, axis=1).ravel()
    cols = np.tile(cells, (1, k)).ravel()
    vals = (area[:, None, None]*m_loc[None]).ravel()
    n = mesh.num_vertices
    return coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsc()


def project_to_p1(mesh, integrand):
    """L2 projection onto CG-P1 of ``integrand(lam, cells)`` evaluated per quadrature point (barycentric ``lam``)."""
    bary, w = cell_quadrature(mesh.cells.shape[1])
    area = mesh.cell_areas()
    cells = mesh.cells
    b = np.zeros(mesh.num_vertices)
    for lam, wq in zip(bary, w):
        val = integrand(lam, cells)
        for i in range(cells.shape[1]):
            np.add.at(b, cells[:, i], wq*area*lam[i]*val)
    return spsolve(_p1_mass(mesh), b)


def elem_size_p1(mesh):
    """``get_horizontal_elem_size_2d`` (utility.py:620-640): P1 projection of sqrt(CellVolume)."""
    size = np.sqrt(mesh.cell_areas())
    return project_to_p1(mesh, lambda lam, cells: size)
