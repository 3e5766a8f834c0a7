"""
Minimal ``FunctionSpace`` / ``Function`` stand-ins for the pieces of Firedrake the 2D driver touches:
P1 (CG, vertex values), P1DG (cell-node values) and their vector versions, ``interpolate`` / ``project`` / ``assign``.

Host arrays use the layout the C ABI takes (include/swe2d.h): DG node 3c+i = vertex i of cell c.
Expressions are callables ``f(x, y)`` (UFL is not available), numbers, ``Constant``s or other ``Function``s.
"""
import numpy as np

from .options import Constant

__all__ = ['FunctionSpace', 'Function', 'MixedFunction', 'get_functionspace', 'triangle_quadrature',
           'quadrilateral_quadrature', 'cell_quadrature']

# 6-point, degree-4 Dunavant rule (barycentric points, weights sum to 1)
_a1, _b1, _w1 = 0.445948490915965, 0.108103018168070, 0.223381589678011
_a2, _b2, _w2 = 0.091576213509771, 0.816847572980459, 0.109951743655322


def triangle_quadrature():
    bary = np.array([[_b1, _a1, _a1], [_a1, _b1, _a1], [_a1, _a1, _b1],
                     [_b2, _a2, _a2], [_a2, _b2, _a2], [_a2, _a2, _b2]])
    w = np.array([_w1, _w1, _w1, _w2, _w2, _w2])
    return bary, w/w.sum()


def quadrilateral_quadrature(n=2):
    """Tensor Gauss-Legendre rule on the unit square: basis values phi (n*n, 4) (nodes counter-clockwise from (0,0))
    and weights summing to 1."""
    gx, gw = np.polynomial.legendre.leggauss(n)
    gx, gw = 0.5*(gx + 1.0), 0.5*gw
    phi, w = [], []
    for xi, wx in zip(gx, gw):
        for ze, wz in zip(gx, gw):
            phi.append([(1 - xi)*(1 - ze), xi*(1 - ze), xi*ze, (1 - xi)*ze])
            w.append(wx*wz)
    return np.array(phi), np.array(w)


def cell_quadrature(npc):
    return triangle_quadrature() if npc == 3 else quadrilateral_quadrature()


class FunctionSpace(object):
    def __init__(self, mesh, family, degree, vector=False, name=None):
        family = {'DQ': 'DG', 'Discontinuous Lagrange': 'DG', 'Lagrange': 'CG'}.get(family, family)
        if family not in ('CG', 'DG') or degree not in (0, 1):
            raise NotImplementedError('only CG1, DG0 and DG1 spaces exist on this path')
        if family == 'CG' and degree == 0:
            raise ValueError('CG0 does not exist')
        self.mesh_obj, self.family, self.degree, self.vector, self.name = mesh, family, degree, vector, name

    def mesh(self):
        return self.mesh_obj

    @property
    def npc(self):
        return int(self.mesh_obj.cells.shape[1])

    def dim(self):
        return self.node_count()*(2 if self.vector else 1)

    def node_count(self):
        m = self.mesh_obj
        return m.num_vertices if self.family == 'CG' else m.num_cells*(self.npc if self.degree == 1 else 1)

    def node_xy(self):
        m = self.mesh_obj
        if self.family == 'CG':
            return m.vertex_xy
        if self.degree == 1:
            return m.cell_xy().reshape(-1, 2)
        return m.cell_xy().mean(axis=1)


def get_functionspace(mesh, h_family, h_degree, v_family=None, v_degree=None, vector=False, name=None, **kwargs):
    """thetis/utility.py:135-189 restricted to 2D meshes."""
    return FunctionSpace(mesh, h_family, h_degree, vector=vector, name=name)


class _Dat(object):
    def __init__(self, func):
        self._f = func

    @property
    def data(self):
        self._f._pull()
        self._f._host_version += 1      # a writable view was handed out
        return self._f._data

    @property
    def data_ro(self):
        self._f._pull()
        return self._f._data


class Function(object):
    def __init__(self, function_space, name=None, val=None):
        self._fs = function_space
        self._name = name
        shape = (function_space.node_count(), 2) if function_space.vector else (function_space.node_count(),)
        self._data = np.zeros(shape)
        self._host_version = 0
        self._pull_hook = None          # set by a device time stepper: refreshes _data from HBM when stale
        self.dat = _Dat(self)
        if val is not None:
            self.assign(val)

    def function_space(self):
        return self._fs

    def name(self):
        return self._name

    def _pull(self):
        if self._pull_hook is not None:
            self._pull_hook()

    # ---- writes
    def assign(self, value):
        self._pull()
        if isinstance(value, Function):
            if value._fs.node_count() == self._fs.node_count() and value._fs.vector == self._fs.vector:
                self._data[...] = value.dat.data_ro
            else:
                self._data[...] = value._as_space(self._fs)
        elif isinstance(value, Constant):
            self._data[...] = np.asarray(value.values() if self._fs.vector else float(value))
        else:
            self._data[...] = np.asarray(value, dtype=float)
        self._host_version += 1
        return self

    # -- a Function inside an expression (thetis_amd/expr.py): lazy, evaluated on the nodes it lives on
    def _expr(self):
        from .expr import Expr, _value
        return Expr(lambda x, y: _value(self, x, y))

    def __add__(self, o): return self._expr() + o
    def __radd__(self, o): return o + self._expr()
    def __sub__(self, o): return self._expr() - o
    def __rsub__(self, o): return o - self._expr()
    def __mul__(self, o): return self._expr()*o
    def __rmul__(self, o): return o*self._expr()
    def __truediv__(self, o): return self._expr()/o
    def __rtruediv__(self, o): return o/self._expr()
    def __pow__(self, o): return self._expr()**o
    def __neg__(self): return -self._expr()

    def interpolate(self, expr):
        """Nodal interpolation."""
        from .expr import evaluation_points
        self._pull()
        with evaluation_points('nodes', self._fs):
            self._data[...] = _evaluate(expr, self._fs.node_xy(), self._fs)
        self._host_version += 1
        return self

    def project(self, expr):
        """L2 projection (``elev_2d.project(elev)``, solver2d.py:763-766).  Into DG-P1 this is a cell-local 3x3 solve;
        CG-P1 -> DG-P1 is exact nodal injection."""
        fs = self._fs
        if isinstance(expr, Function) or isinstance(expr, Constant) or np.isscalar(expr) \
                or (not callable(expr)):
            # P1 (CG or DG on the same mesh) and constants are in the DG-P1 space: projection = injection
            return self.interpolate(expr)
        if fs.family == 'CG' and fs.degree == 1 and not fs.vector:
            # global L2 projection onto the continuous P1 space (bathymetry_2d.project(expr) in the reference's tests)
            from .cgproject import project_to_p1
            mesh = fs.mesh_obj
            p = mesh.cell_xy()

            def integrand(lam, cells):
                from .expr import evaluation_points
                xq = np.einsum('nic,i->nc', p, lam)
                with evaluation_points('cells', mesh, lam):
                    return np.asarray(expr(xq[:, 0], xq[:, 1]))*np.ones(len(cells))
            self._pull()
            self._data[...] = project_to_p1(mesh, integrand).reshape(self._data.shape)
            self._host_version += 1
            return self
        if fs.family != 'DG' or fs.degree != 1:
            raise NotImplementedError('projection of expressions is implemented for DG-P1 and scalar CG-P1 targets only')
        mesh = fs.mesh_obj
        npc = fs.npc
        bary, w = cell_quadrature(npc)
        p = mesh.cell_xy()
        n = mesh.num_cells
        ncomp = 2 if fs.vector else 1
        from .expr import evaluation_points
        b = np.zeros((n, npc, ncomp))
        for l, wq in zip(bary, w):
            xq = np.einsum('nic,i->nc', p, l)
            with evaluation_points('cells', mesh, l):
                val = expr(xq[:, 0], xq[:, 1])
            if fs.vector:
                val = np.stack([np.asarray(val[0])*np.ones(n), np.asarray(val[1])*np.ones(n)], axis=1)
            else:
                val = (np.asarray(val)*np.ones(n))[:, None]
            for i in range(npc):
                b[:, i, :] += wq*l[i]*val          # divided by the cell area
        if npc == 3:
            # (M/A)^-1 = 12 [[2,1,1],..]^-1  ->  x_i = 3 (4 b_i - sum b)
            x = 3.0*(4.0*b - b.sum(axis=1, keepdims=True))
        elif getattr(mesh, 'affine', True):
            # parallelogram: (M/A)^-1 = m^-1 (x) m^-1, m^-1 = [[4,-2],[-2,4]]
            x = 16.0*b - 8.0*np.roll(b, -1, axis=1) - 8.0*np.roll(b, 1, axis=1) + 4.0*np.roll(b, 2, axis=1)
        else:
            # general quadrilateral: det J = d0 + d1 xi + d2 zeta varies over the cell - weight the right-hand side with it and
            # solve with the true 4 x 4 mass matrix of every cell (same 2 x 2 rule: exact for M)
            a, bb = p[:, 1] - p[:, 0], p[:, 3] - p[:, 0]
            c = p[:, 0] - p[:, 1] + p[:, 2] - p[:, 3]
            cross = lambda u, v: u[:, 0]*v[:, 1] - u[:, 1]*v[:, 0]
            d0, d1, d2 = cross(a, bb), cross(a, c), cross(c, bb)
            b[...] = 0.0
            M = np.zeros((n, 4, 4))
            for l, wq in zip(bary, w):
                xi, ze = l[1] + l[2], l[2] + l[3]
                det = wq*(d0 + d1*xi + d2*ze)
                xq = np.einsum('nic,i->nc', p, l)
                with evaluation_points('cells', mesh, l):
                    val = expr(xq[:, 0], xq[:, 1])
                if fs.vector:
                    val = np.stack([np.asarray(val[0])*np.ones(n), np.asarray(val[1])*np.ones(n)], axis=1)
                else:
                    val = (np.asarray(val)*np.ones(n))[:, None]
                b += det[:, None, None]*l[None, :, None]*val[:, None, :]
                M += det[:, None, None]*np.outer(l, l)[None]
            x = np.linalg.solve(M, b)
        self._pull()
        self._data[...] = x.reshape(self._data.shape)
        self._host_version += 1
        return self

    # ---- reads
    def _as_space(self, fs):
        """Values of this P1 function at the nodes of another P1 space on the same mesh."""
        mesh = fs.mesh_obj
        src = self._fs
        data = self.dat.data_ro
        if src.family == 'CG' and fs.family == 'DG' and fs.degree == 1:
            return data[mesh.cells.reshape(-1)]
        if src.family == 'DG' and src.degree == 0 and fs.family == 'DG' and fs.degree == 1:
            return np.repeat(data, fs.npc, axis=0)
        raise NotImplementedError('cannot inject {:}{:} into {:}{:}'.format(src.family, src.degree, fs.family, fs.degree))

    def cell_node_values(self):
        """(N, k[, 2]) values at the nodes of every cell."""
        mesh = self._fs.mesh_obj
        d = self.dat.data_ro
        k = self._fs.npc
        if self._fs.family == 'CG':
            return d[mesh.cells]
        if self._fs.degree == 1:
            return d.reshape((mesh.num_cells, k) + d.shape[1:])
        return np.repeat(d[:, None], k, axis=1)

    def at(self, xy):
        raise NotImplementedError('point evaluation is not part of the hot path')


def _evaluate(expr, xy, fs):
    n = xy.shape[0]
    if isinstance(expr, Function):
        return expr._as_space(fs) if expr._fs.node_count() != n or expr._fs.family != fs.family else expr.dat.data_ro
    if isinstance(expr, Constant):
        return np.asarray(expr.values())*np.ones((n, 2)) if fs.vector else float(expr)*np.ones(n)
    if callable(expr):
        val = expr(xy[:, 0], xy[:, 1])
        if fs.vector:
            return np.stack([np.asarray(val[0])*np.ones(n), np.asarray(val[1])*np.ones(n)], axis=1)
        return np.asarray(val)*np.ones(n)
    val = np.asarray(expr, dtype=float)
    return val*np.ones((n, 2)) if fs.vector else val*np.ones(n)


class MixedFunction(object):
    """``solution_2d`` on V_2d = U_2d x H_2d (solver2d.py:345, 410-413)."""

    def __init__(self, subfunctions, name=None):
        self.subfunctions = tuple(subfunctions)
        self._name = name

    def split(self):
        return self.subfunctions

    def assign(self, other):
        for a, b in zip(self.subfunctions, other.subfunctions):
            a.assign(b)
        return self
