"""
2D meshes as flat arrays (the host-side stand-in for Firedrake's ``RectangleMesh`` & co).

The reference never touches mesh arrays itself - it calls ``RectangleMesh(nx, ny, lx, ly)``
(examples/channel2d/channel2d.py:25), ``PeriodicRectangleMesh`` (test/swe2d/test_rossby_wave.py)
or ``UnitSquareMesh(.., quadrilateral=True)`` (demos/demo_2d_tracer.py:19) and lets Firedrake/DMPlex
own topology.  Here a mesh is a handful of numpy arrays that go straight to HBM:

``vertex_xy``   (V, 2) float64   geometric vertices (periodic meshes keep *unwrapped* duplicates)
``cells``       (N, k) int32     counter-clockwise vertex ids of every triangle (k=3) or quadrilateral (k=4)
``cell_nbr``    (N, k) int32     facet f joins local vertices f and (f+1)%k;  >=0: neighbour cell,
                                 <0: ``-marker`` of the boundary the facet lies on
``cell_nbr_facet`` (N, k) int8   local facet number of the same facet inside the neighbour

Quadrilaterals: parallelograms take the affine kernels (every quadrilateral mesh the reference builds itself is a rectangle
grid); any other convex quadrilateral the general bilinear ones.

Conventions [FD-assumed, SURVEY.md A.8]: vertices on a regular grid, 'left' diagonal
(from (i, j+1) to (i+1, j)), boundary markers 1: x=0, 2: x=Lx, 3: y=0, 4: y=Ly.
"""
import numpy as np

__all__ = ['Mesh2d', 'RectangleMesh', 'PeriodicRectangleMesh', 'UnitSquareMesh', 'SquareMesh']


def _signed_area2(p):
    """Twice the signed area of polygons p (N, k, 2)."""
    d = p[:, 1:] - p[:, :1]                 # fan from vertex 0: differences first, no cancellation far from the origin
    return np.sum(d[:, :-1, 0]*d[:, 1:, 1] - d[:, 1:, 0]*d[:, :-1, 1], axis=1)


class Mesh2d(object):
    """2D mesh of triangles or (convex) quadrilaterals with facet-neighbour connectivity."""

    def __init__(self, vertex_xy, cells, topo_vertex=None, marker_fn=None, name='mesh2d'):
        """
        :arg vertex_xy: (V, 2) vertex coordinates
        :arg cells: (N, 3) vertex ids (any orientation; made counter-clockwise here)
        :kwarg topo_vertex: (V,) canonical vertex id used for *topology* (periodic identification)
        :kwarg marker_fn: callable(xm, ym) -> int marker (>0) for exterior facet mid-points
        """
        self.name = name
        self.vertex_xy = np.ascontiguousarray(vertex_xy, dtype=np.float64)
        cells = np.array(cells, dtype=np.int32, copy=True)
        assert cells.ndim == 2 and cells.shape[1] in (3, 4), 'cells must be triangles or quadrilaterals'
        self.nodes_per_cell = int(cells.shape[1])
        # orient counter-clockwise
        p = self.vertex_xy[cells]
        area2 = _signed_area2(p)
        if np.any(area2 == 0):
            raise ValueError('degenerate (zero area) cell in mesh')
        flip = area2 < 0
        self.affine = True
        if self.nodes_per_cell == 3:
            cells[flip, 1], cells[flip, 2] = cells[flip, 2].copy(), cells[flip, 1].copy()
        else:
            cells[flip, 1], cells[flip, 3] = cells[flip, 3].copy(), cells[flip, 1].copy()
            p = self.vertex_xy[cells]
            c = p[:, 0] - p[:, 1] + p[:, 2] - p[:, 3]
            # a cell that is not a parallelogram selects the general bilinear kernels (csrc/swe2d_kernels.h: swe_quad_mass;
            # solver2d.py:340-345 of the reference accepts any quadrilateral mesh); it must be convex: det J > 0 at the corners
            self.affine = not np.any(np.abs(c).max(axis=1) > 1e-9*np.sqrt(np.abs(area2)))
            if not self.affine:
                a, b = p[:, 1] - p[:, 0], p[:, 3] - p[:, 0]
                d0 = a[:, 0]*b[:, 1] - a[:, 1]*b[:, 0]
                d1 = a[:, 0]*c[:, 1] - a[:, 1]*c[:, 0]
                d2 = c[:, 0]*b[:, 1] - c[:, 1]*b[:, 0]
                if np.any(np.minimum.reduce([d0, d0 + d1, d0 + d2, d0 + d1 + d2]) <= 0.0):
                    raise ValueError('quadrilateral cells must be convex')
        self.cells = np.ascontiguousarray(cells)
        self.topo_vertex = (np.arange(len(self.vertex_xy), dtype=np.int64) if topo_vertex is None
                            else np.asarray(topo_vertex, dtype=np.int64))
        self._build_connectivity(marker_fn)
        self.boundary_len = self._boundary_length()

    # ------------------------------------------------------------------ topology
    def _build_connectivity(self, marker_fn):
        n = self.num_cells
        tv = self.topo_vertex[self.cells]                       # (N, 3) canonical ids
        a = tv                                                  # facet f: a -> b
        b = np.roll(tv, -1, axis=1)
        lo = np.minimum(a, b).ravel()
        hi = np.maximum(a, b).ravel()
        nv = int(self.topo_vertex.max()) + 1
        key = lo*nv + hi
        order = np.argsort(key, kind='stable')
        ks = key[order]
        same_next = np.zeros(len(ks), dtype=bool)
        same_next[:-1] = ks[1:] == ks[:-1]
        same_prev = np.zeros(len(ks), dtype=bool)
        same_prev[1:] = same_next[:-1]
        if np.any(same_next & same_prev):
            raise ValueError('non-manifold mesh: a facet is shared by more than two cells')
        k = self.nodes_per_cell
        nbr = np.full(k*n, np.iinfo(np.int32).min, dtype=np.int64)
        nbf = np.zeros(k*n, dtype=np.int8)
        first = order[same_next]
        second = order[same_prev]
        nbr[first] = second // k
        nbf[first] = second % k
        nbr[second] = first // k
        nbf[second] = first % k
        ext = order[~(same_next | same_prev)]
        if len(ext):
            c, f = ext // k, ext % k
            pa = self.vertex_xy[self.cells[c, f]]
            pb = self.vertex_xy[self.cells[c, (f + 1) % k]]
            mid = 0.5*(pa + pb)
            if marker_fn is None:
                markers = np.ones(len(ext), dtype=np.int64)
            else:
                markers = np.asarray(marker_fn(mid[:, 0], mid[:, 1]), dtype=np.int64)
            if np.any(markers <= 0):
                raise ValueError('boundary markers must be positive integers')
            nbr[ext] = -markers
        self.cell_nbr = np.ascontiguousarray(nbr.reshape(n, k).astype(np.int32))
        self.cell_nbr_facet = np.ascontiguousarray(nbf.reshape(n, k))

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def num_vertices(self):
        return self.vertex_xy.shape[0]

    @property
    def boundary_markers(self):
        # cached per connectivity array: the time steppers ask for it at every stage (update_forcings)
        cache = self.__dict__.get('_boundary_markers')
        if cache is None or cache[0] is not self.cell_nbr:
            m = -self.cell_nbr[self.cell_nbr < 0]
            cache = (self.cell_nbr, sorted(int(i) for i in np.unique(m)))
            self.__dict__['_boundary_markers'] = cache
        return list(cache[1])

    def cell_xy(self):
        """(N, k, 2) coordinates of the DG-P1 / DQ-1 nodes (= cell vertices)."""
        return self.vertex_xy[self.cells]

    def cell_areas(self):
        return 0.5*_signed_area2(self.cell_xy())

    def _boundary_length(self):
        """Total length of every boundary marker (thetis/utility.py:821-832, ``assemble(1*ds(i))``)."""
        out = {}
        c, f = np.nonzero(self.cell_nbr < 0)
        pa = self.vertex_xy[self.cells[c, f]]
        pb = self.vertex_xy[self.cells[c, (f + 1) % self.nodes_per_cell]]
        ln = np.hypot(*(pb - pa).T)
        mk = -self.cell_nbr[c, f]
        for m in np.unique(mk):
            out[int(m)] = float(ln[mk == m].sum())
        return out

    # ------------------------------------------------------------------ renumbering
    def renumbered(self, perm):
        """Return a copy whose cell ``i`` is this mesh's cell ``perm[i]`` (vertices untouched)."""
        perm = np.asarray(perm, dtype=np.int64)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        new = object.__new__(Mesh2d)
        new.name = self.name
        new.nodes_per_cell = self.nodes_per_cell
        new.vertex_xy = self.vertex_xy
        new.topo_vertex = self.topo_vertex
        new.cells = np.ascontiguousarray(self.cells[perm])
        nb = self.cell_nbr[perm].astype(np.int64)
        pos = nb >= 0
        nb[pos] = inv[nb[pos]]
        new.cell_nbr = np.ascontiguousarray(nb.astype(np.int32))
        new.cell_nbr_facet = np.ascontiguousarray(self.cell_nbr_facet[perm])
        new.boundary_len = dict(self.boundary_len)
        for k in ('nx', 'ny', 'lx', 'ly'):
            if hasattr(self, k):
                setattr(new, k, getattr(self, k))
        return new


def _grid_cells(nx, ny, diagonal):
    """Triangles of an (nx x ny) grid of quads, vertex id = i*(ny+1) + j, quad-major ordering in x."""
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing='xy')     # j-major: rows of constant j
    i = i.ravel()
    j = j.ravel()
    v00 = i*(ny + 1) + j
    v01 = i*(ny + 1) + j + 1
    v11 = (i + 1)*(ny + 1) + j + 1
    v10 = (i + 1)*(ny + 1) + j
    if diagonal == 'left':
        t0 = np.stack([v00, v10, v01], axis=1)
        t1 = np.stack([v01, v10, v11], axis=1)
    elif diagonal == 'right':
        t0 = np.stack([v00, v10, v11], axis=1)
        t1 = np.stack([v00, v11, v01], axis=1)
    else:
        raise ValueError('diagonal must be "left" or "right"')
    cells = np.empty((2*nx*ny, 3), dtype=np.int64)
    cells[0::2] = t0
    cells[1::2] = t1
    return cells


def _grid_quads(nx, ny):
    """Quadrilaterals of an (nx x ny) grid, cell = j*nx + i, vertices counter-clockwise from (i, j)."""
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing='xy')
    i = i.ravel()
    j = j.ravel()
    return np.stack([i*(ny + 1) + j, (i + 1)*(ny + 1) + j, (i + 1)*(ny + 1) + j + 1, i*(ny + 1) + j + 1], axis=1)


def _rect_marker_fn(lx, ly, periodic_x=False, periodic_y=False):
    def fn(xm, ym):
        tol = 1e-9*max(lx, ly)
        m = np.zeros(xm.shape, dtype=np.int64)
        m[np.abs(xm) < tol] = 1
        m[np.abs(xm - lx) < tol] = 2
        m[np.abs(ym) < tol] = 3
        m[np.abs(ym - ly) < tol] = 4
        return m
    return fn


def RectangleMesh(nx, ny, lx, ly, quadrilateral=False, diagonal='left', name='mesh2d'):
    """``RectangleMesh(nx, ny, Lx, Ly)``: 2*nx*ny triangles or nx*ny quadrilaterals, markers 1..4 [FD-assumed, A.8]."""
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)
    xx, yy = np.meshgrid(xs, ys, indexing='ij')
    vertex_xy = np.stack([xx.ravel(), yy.ravel()], axis=1)
    cells = _grid_quads(nx, ny) if quadrilateral else _grid_cells(nx, ny, diagonal)
    mesh = Mesh2d(vertex_xy, cells, marker_fn=_rect_marker_fn(lx, ly), name=name)
    mesh.nx, mesh.ny, mesh.lx, mesh.ly = nx, ny, float(lx), float(ly)
    mesh.structured = True              # cell = 2*(j*nx + i) + t: lets the device pick a tiled numbering
    return mesh


def PeriodicRectangleMesh(nx, ny, lx, ly, direction='x', quadrilateral=False, diagonal='left', name='mesh2d'):
    """Rectangle periodic in ``direction`` ('x', 'y' or 'both'); geometry stays unwrapped."""
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)
    xx, yy = np.meshgrid(xs, ys, indexing='ij')
    vertex_xy = np.stack([xx.ravel(), yy.ravel()], axis=1)
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing='ij')
    if direction in ('x', 'both'):
        ii = ii % nx
    if direction in ('y', 'both'):
        jj = jj % ny
    topo = (ii*(ny + 1) + jj).ravel()
    cells = _grid_quads(nx, ny) if quadrilateral else _grid_cells(nx, ny, diagonal)
    mesh = Mesh2d(vertex_xy, cells, topo_vertex=topo, marker_fn=_rect_marker_fn(lx, ly), name=name)
    mesh.nx, mesh.ny, mesh.lx, mesh.ly = nx, ny, float(lx), float(ly)
    mesh.structured = True
    return mesh


def SquareMesh(nx, ny, l, **kwargs):
    return RectangleMesh(nx, ny, l, l, **kwargs)


def UnitSquareMesh(nx, ny, **kwargs):
    return RectangleMesh(nx, ny, 1.0, 1.0, **kwargs)
