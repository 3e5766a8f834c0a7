"""
Firedrake-shaped mesh arrays -> the arrays of ``swe2d_mesh`` (include/swe2d.h).  Pure numpy: this is the part of the
reference-side binding that can be pinned without Firedrake - the array contract.

What a Thetis maintainer passes in, straight from a Firedrake ``mesh`` (all [FD-assumed]: Firedrake cannot be imported here):

``coords``          ``mesh.coordinates.dat.data_ro``                       (V, 2)  CG-P1 coordinate dofs = vertices
``cell_vertices``   ``mesh.coordinates.cell_node_map().values``            (N, 3 | 4) vertex ids per cell, FIAT vertex order,
                                                                                   either orientation
``int_facet_cell``  ``mesh.interior_facets.facet_cell``                    (Fi, 2) the two cells of every interior facet
``int_local_facet`` ``mesh.interior_facets.local_facet_dat.data_ro``       (Fi, 2) its local facet number in each of them
``ext_facet_cell``  ``mesh.exterior_facets.facet_cell``                    (Fe, 1) or (Fe,)
``ext_local_facet`` ``mesh.exterior_facets.local_facet_dat.data_ro``       (Fe,)
``ext_markers``     ``mesh.exterior_facets.markers``                       (Fe,)   ``ds(marker)`` ids
``dg_cell_nodes``   ``FunctionSpace(mesh, 'DG', 1).cell_node_map().values`` (N, 3) dof ids of ``uv_2d`` / ``elev_2d`` per cell

FIAT / UFC simplex numbering: local facet i of a triangle is the edge OPPOSITE local vertex i.  Quadrilaterals are FIAT
tensor-product cells: local vertices in lexicographic order of the reference coordinates, 0:(0,0) 1:(0,1) 2:(1,0) 3:(1,1) (NOT
cyclic), local facets 0: x = 0 {0,1}, 1: x = 1 {2,3}, 2: y = 0 {0,2}, 3: y = 1 {1,3}; DQ-1 dofs in the same lexicographic order.
This library numbers facet f as the edge from local vertex f to f + 1 of a counter-clockwise cell (vertices in cyclic order).  Partitioned runs (one MPI rank <-> one handle): pass the
rank's local arrays (owned + ghost cells, Firedrake's local numbering); facets of ghost cells that have no local neighbour
arrive as exterior facets without a marker - give them ``halo_marker`` (they are never updated, see thetis_amd/partition.py).
"""
import numpy as np

__all__ = ['swe2d_mesh_arrays']


def swe2d_mesh_arrays(coords, cell_vertices, int_facet_cell, int_local_facet, ext_facet_cell, ext_local_facet, ext_markers,
                      dg_cell_nodes=None, halo_marker=None):
    """Returns a dict with ``vertex_xy`` (V,2), ``cell_vertices`` (N,k, counter-clockwise cyclic order; k = 3 or 4),
    ``cell_neighbours`` (N,k: >= 0 neighbour cell, < 0: -marker), ``cell_neighbour_facets`` (N,k int8) and, when
    ``dg_cell_nodes`` is given, ``dg_perm`` (N,k): ``uv_2d.dat.data[dg_perm.ravel()]`` is the (kN, 2) cell-major array of
    ``swe2d_set_state`` (and
    ``uv_2d.dat.data[dg_perm.ravel()] = ...`` writes a ``swe2d_get_state`` result back)."""
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    cv = np.array(cell_vertices, dtype=np.int64, copy=True)
    n = cv.shape[0]
    if cv.ndim != 2 or cv.shape[1] not in (3, 4):
        raise NotImplementedError('triangles (N, 3) or tensor-product quadrilaterals (N, 4)')
    k = cv.shape[1]
    p = xy[cv]
    if k == 3:
        area2 = (p[:, 1, 0] - p[:, 0, 0])*(p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1])*(p[:, 2, 0] - p[:, 0, 0])
        # position of the old local vertex j in the counter-clockwise cell: clockwise cells swap local vertices 1 and 2
        ccw_pos, cw_pos = np.array([0, 1, 2]), np.array([0, 2, 1])
        facet_vertices = np.array([[1, 2], [2, 0], [0, 1]])                  # FIAT facet i: the edge opposite vertex i
    else:
        # the cycle through the lexicographic vertices is 0 -> 2 -> 3 -> 1 (or backwards for a mirrored cell)
        cyc = p[:, [0, 2, 3, 1]]
        nxt = np.roll(cyc, -1, axis=1)
        area2 = (cyc[:, :, 0]*nxt[:, :, 1] - nxt[:, :, 0]*cyc[:, :, 1]).sum(axis=1)
        ccw_pos, cw_pos = np.array([0, 3, 1, 2]), np.array([0, 1, 3, 2])    # old local vertex j -> its place in the cycle
        facet_vertices = np.array([[0, 1], [2, 3], [0, 2], [1, 3]])
    if np.any(area2 == 0):
        raise ValueError('degenerate cell')
    flip = area2 < 0
    pos = np.tile(ccw_pos, (n, 1))
    pos[flip] = cw_pos
    cells = np.empty_like(cv)
    np.put_along_axis(cells, pos, cv, axis=1)

    def our_facet(cell, fiat_facet):
        """FIAT facet i -> the facet f of this library with {f, f + 1} = the new positions of the edge's two vertices"""
        a = np.take_along_axis(pos[cell], facet_vertices[fiat_facet, 0][:, None], axis=1)[:, 0]
        b = np.take_along_axis(pos[cell], facet_vertices[fiat_facet, 1][:, None], axis=1)[:, 0]
        if np.any(((a + 1) % k != b) & ((b + 1) % k != a)):
            raise ValueError('a local facet does not join two consecutive vertices of its cell')
        return np.where((a + 1) % k == b, a, b)

    nbr = np.full((n, k), np.iinfo(np.int32).min, dtype=np.int64)
    nbf = np.zeros((n, k), dtype=np.int8)
    ifc = np.asarray(int_facet_cell, dtype=np.int64).reshape(-1, 2)
    ilf = np.asarray(int_local_facet, dtype=np.int64).reshape(-1, 2)
    f0, f1 = our_facet(ifc[:, 0], ilf[:, 0]), our_facet(ifc[:, 1], ilf[:, 1])
    nbr[ifc[:, 0], f0], nbf[ifc[:, 0], f0] = ifc[:, 1], f1
    nbr[ifc[:, 1], f1], nbf[ifc[:, 1], f1] = ifc[:, 0], f0
    efc = np.asarray(ext_facet_cell, dtype=np.int64).reshape(-1)
    elf = np.asarray(ext_local_facet, dtype=np.int64).reshape(-1)
    mk = np.asarray(ext_markers, dtype=np.int64).reshape(-1)
    if halo_marker is not None:
        mk = np.where(mk <= 0, int(halo_marker), mk)
    if np.any(mk <= 0):
        raise ValueError('exterior facet without a positive marker (pass halo_marker for the unmarked facets of ghost cells)')
    fe = our_facet(efc, elf)
    nbr[efc, fe] = -mk
    if np.any(nbr == np.iinfo(np.int32).min):
        raise ValueError('a cell facet is neither in the interior nor in the exterior facet set')
    out = {'vertex_xy': xy, 'cell_vertices': np.ascontiguousarray(cells, dtype=np.int32),
           'cell_neighbours': np.ascontiguousarray(nbr, dtype=np.int32), 'cell_neighbour_facets': nbf}
    if dg_cell_nodes is not None:
        dg = np.asarray(dg_cell_nodes, dtype=np.int64)
        perm = np.empty_like(dg)
        np.put_along_axis(perm, pos, dg, axis=1)          # DG-P1 / DQ-1 'equispaced' nodes sit on the vertices, in the cell's vertex order
        out['dg_perm'] = perm
    return out
