"""
thetis_amd - MI355X-native explicit 2D shallow-water time stepper behind Thetis's FlowSolver2d surface.

    from thetis_amd import *
    mesh2d = RectangleMesh(80, 3, 100e3, 3750)
    bathymetry_2d = Function(get_functionspace(mesh2d, 'CG', 1)).interpolate(lambda x, y: 20 - 15*x/100e3)
    solver_obj = solver2d.FlowSolver2d(mesh2d, bathymetry_2d)
    solver_obj.options.swe_timestepper_type = 'SSPRK33'
    ...
    solver_obj.iterate()

Importing this package never loads the HIP extension; creating a time stepper does, and fails loudly without it.

Several GPUs: the same script under ``python -m torch.distributed.run --nproc-per-node N script.py`` (one process per GPU) is
domain-decomposed as ``mpiexec -n N`` does it for the reference (thetis_amd/comm.py, thetis_amd/spmd.py).
"""
import os  # noqa: F401  (the reference's scripts use os.getenv after `from thetis import *`)

from . import solver2d  # noqa: F401
from .expr import *  # noqa: F401,F403  (SpatialCoordinate, conditional, as_vector, sin, cos, exp, sqrt, pi ...)
from .function import Function, FunctionSpace, get_functionspace  # noqa: F401
from .mesh import Mesh2d, PeriodicRectangleMesh, RectangleMesh, SquareMesh, UnitSquareMesh  # noqa: F401
from .meshio import read_gmsh, write_gmsh  # noqa: F401
from .options import Constant, ModelOptions2d  # noqa: F401
from .shallowwater_eq import g_grav, physical_constants, rho_0  # noqa: F401


def Mesh(path, **kwargs):
    """``Mesh('file.msh')`` as in Firedrake user scripts: Gmsh MSH 2.2 files."""
    return read_gmsh(path, **kwargs)
