"""
CG-P1 L2 projections used only to pick the automatic CFL time step at setup
(thetis/solver2d.py:149-177, thetis/utility.py:620-640): ``solve(inner(test, trial)*dx == inner(test, f)*dx)`` on the
continuous P1 space.  Host-side (scipy sparse), runs once.
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.linalg import spsolve

from .function import triangle_quadrature


def _p1_mass(mesh):
    cells = mesh.cells
    area = mesh.cell_areas()
    m_loc = np.array([[2.0, 1.0, 1.0], [1.0, 2.0, 1.0], [1.0, 1.0, 2.0]])/12.0
    rows = np.repeat(cells, 3, axis=1).ravel()
    cols = np.tile(cells, (1, 3)).ravel()
    vals = (area[:, None, None]*m_loc[None]).ravel()
    n = mesh.num_vertices
    return coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsc()


def project_to_p1(mesh, integrand):
    """L2 projection onto CG-P1 of ``integrand(lam, cells)`` evaluated per quadrature point (barycentric ``lam``)."""
    bary, w = triangle_quadrature()
    area = mesh.cell_areas()
    cells = mesh.cells
    b = np.zeros(mesh.num_vertices)
    for lam, wq in zip(bary, w):
        val = integrand(lam, cells)
        for i in range(3):
            np.add.at(b, cells[:, i], wq*area*lam[i]*val)
    return spsolve(_p1_mass(mesh), b)


def elem_size_p1(mesh):
    """``get_horizontal_elem_size_2d`` (utility.py:620-640): P1 projection of sqrt(CellVolume)."""
    size = np.sqrt(mesh.cell_areas())
    return project_to_p1(mesh, lambda lam, cells: size)
