"""
Locality ordering of cells and vertices for the device (invisible to callers: ``Swe2dDevice`` permutes on the way in
and out).  Consecutive cells should be mesh neighbours so that (i) a 256-cell workgroup is a compact patch whose
facet-neighbour gathers mostly hit lines it loaded itself and (ii) the contiguous eighth of the cell range that every
XCD works on (swe_logical_block in csrc/swe2d_kernels.h) only shares a thin seam with the other XCDs' L2s.
Firedrake gives the reference the same service by reordering DMPlex points (reverse Cuthill-McKee) [FD-assumed].
"""
import numpy as np

__all__ = ['hilbert_index', 'hilbert_cell_order', 'tile_cell_order', 'structured_tile_order', 'structured_subset_order', 'auto_cell_order', 'flow_block_order', 'fused_tile_order', 'triple_tile_order', 'bisection_block_order',
           'patch_row_order', 'first_touch_vertex_order']


def hilbert_index(ix, iy, order):
    """Hilbert curve index of integer grid points (ix, iy) in [0, 2^order)^2 (vectorised xy -> d)."""
    ix = ix.astype(np.int64).copy()
    iy = iy.astype(np.int64).copy()
    d = np.zeros_like(ix)
    s = 1 << (order - 1)
    while s > 0:
        rx = (ix & s) > 0
        ry = (iy & s) > 0
        d += s*s*((3*rx.astype(np.int64)) ^ ry.astype(np.int64))
        # rotate
        flip = ~ry & rx
        ix = np.where(flip, s - 1 - ix, ix)
        iy = np.where(flip, s - 1 - iy, iy)
        swap = ~ry
        ix, iy = np.where(swap, iy, ix), np.where(swap, ix, iy)
        s >>= 1
    return d


def hilbert_cell_order(cell_centroids, order=16):
    """Permutation ``perm`` such that new cell i = old cell perm[i], sorted along a Hilbert curve through the centroids."""
    c = np.asarray(cell_centroids, dtype=np.float64)
    lo = c.min(axis=0)
    span = np.maximum(c.max(axis=0) - lo, 1e-300)
    scale = (2**order - 1)/span.max()               # isotropic: keeps the curve's locality on elongated domains
    q = np.floor((c - lo)*scale).astype(np.int64)
    d = hilbert_index(q[:, 0], q[:, 1], order)
    return np.lexsort((np.arange(len(d)), d))


def patch_row_order(cell_centroids, perm, patch=256):
    """Refine a locality order: inside every block of ``patch`` consecutive cells sort by rows (y band, then x) so that
    consecutive lanes of a wave walk along mesh rows (their neighbour gathers then coalesce too)."""
    c = np.asarray(cell_centroids)[perm]
    n = len(perm)
    out = perm.copy()
    for a in range(0, n, patch):
        b = min(a + patch, n)
        cc = c[a:b]
        span = cc.max(axis=0) - cc.min(axis=0)
        # band height ~ typical cell size: area of the patch / number of cells
        hcell = np.sqrt(max(span[0]*span[1], 1e-300)/(b - a))*1.4142
        band = np.floor((cc[:, 1] - cc[:, 1].min())/max(hcell, 1e-300) + 1e-9)
        out[a:b] = perm[a:b][np.lexsort((cc[:, 0], band))]
    return out


def tile_cell_order(mesh, bx=16, by=8):
    """Structured meshes (RectangleMesh): order by tiles of bx x by quads, row-major inside a tile."""
    nx = mesh.nx
    q = np.arange(mesh.num_cells)//2
    i, j = q % nx, q//nx
    return np.lexsort((np.arange(mesh.num_cells), i % bx, j % by, i//bx, j//by))


def structured_tile_order(nx, ny, bx=16, by=8, cells_per_quad=2):
    """RectangleMesh numbering (cell = 2*(j*nx + i) + t, or j*nx + i for quadrilaterals): tiles of bx x by quads visited
    along a Hilbert curve over the tile grid, row-major inside a tile - a 256-cell workgroup is one 16 x 8 tile of
    triangles (16 x 16 of quadrilaterals) and a wave walks along mesh rows."""
    n = cells_per_quad*nx*ny
    q = np.arange(n)//cells_per_quad
    i, j = q % nx, q//nx
    order = max(1, int(np.ceil(np.log2(max(nx//bx + 1, ny//by + 1)))))
    d = hilbert_index(i//bx, j//by, order)
    return np.lexsort((np.arange(n), i % bx, j % by, d))


def structured_subset_order(global_cells, nx, ny, bx=16, by=8, cells_per_quad=2):
    """The tile order of ``structured_tile_order`` restricted to a subset of the cells of a RectangleMesh (the owned or ghost
    cells of a partition, by their global ids): returns positions into ``global_cells``."""
    g = np.asarray(global_cells, dtype=np.int64)
    q = g//cells_per_quad
    i, j = q % nx, q//nx
    order = max(1, int(np.ceil(np.log2(max(nx//bx + 1, ny//by + 1)))))
    d = hilbert_index(i//bx, j//by, order)
    return np.lexsort((g, i % bx, j % by, d))


def auto_cell_order(mesh, a=0, b=None):
    """Default device ordering of cells a..b of ``mesh``: structured tiles when the mesh says it is a plain
    RectangleMesh - or a partition of one (``structured_parent`` = (nx, ny) of the parent, ``local_to_global`` its cell ids:
    a rank's cells keep the tile order the whole mesh would get, 15-20 % faster than the Hilbert curve on that mesh) -, a
    Hilbert curve through the centroids otherwise."""
    b = mesh.cells.shape[0] if b is None else b
    k = np.asarray(mesh.cells).shape[1]
    if a == 0 and b == mesh.cells.shape[0] and getattr(mesh, 'structured', False):
        if k == 4:
            # 16 x 12 quadrilaterals = 192 cells: one tile of the fused stage pair on quadrilaterals (csrc/swe2d_fuse.h: 192 interior cells
            # + their ring of 56 in a 256-lane workgroup); rounds 1-5 had 16 x 16
            return structured_tile_order(mesh.nx, mesh.ny, bx=16, by=12, cells_per_quad=1)
        # 16 x 6 quads = 192 triangles: one tile of the fused stage pair (csrc/swe2d_fuse.h: 192 interior cells + their ring of 44
        # in a 256-lane workgroup); the stage kernels run the same in 16 x 6 and in 16 x 8 tiles (112.3 against 111.4-112.8 us per step)
        return structured_tile_order(mesh.nx, mesh.ny, bx=16, by=6)
    parent = getattr(mesh, 'structured_parent', None)
    if parent is not None:
        g = np.asarray(mesh.local_to_global)[a:b]
        if k == 4:
            return structured_subset_order(g, parent[0], parent[1], bx=16, by=12, cells_per_quad=1)
        return structured_subset_order(g, parent[0], parent[1], bx=16, by=6)      # (the whole mesh's 16 x 6-quad tiles, see above)
    cen = np.asarray(mesh.vertex_xy)[np.asarray(mesh.cells)[a:b]].mean(axis=1)
    return hilbert_cell_order(cen)


def bisection_block_order(centroids, block=64):
    """Cells in the order of the leaves of a recursive coordinate bisection whose leaves hold exactly ``block`` cells (the last one
    the remainder): every cut is along the longer side of the box at the count that gives the left part a multiple of ``block``."""
    cen = np.asarray(centroids, dtype=np.float64)
    out = []
    stack = [np.arange(len(cen), dtype=np.int64)]
    while stack:
        idx = stack.pop()
        m = len(idx)
        if m <= block:
            out.append(idx)
            continue
        nl = ((-(-m//block))//2)*block
        c = cen[idx]
        ext = c.max(axis=0) - c.min(axis=0)
        ax = int(ext[1] > ext[0])
        part = np.argpartition(c[:, ax], nl)
        stack.append(idx[part[nl:]])                 # (popped after the left part: leaves come out left to right)
        stack.append(idx[part[:nl]])
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def flow_block_order(mesh, a=0, b=None):
    """Order of cells a..b of ``mesh`` for the BLOCKS of the dataflow kernel (csrc/swe2d_flow.h: 64 consecutive cells = one wave's
    block; a facet between two blocks is a rim facet whose six trace values travel as granules after every stage).  What counts is
    a block's perimeter, not how its lanes walk through memory - the kernel touches the state planes at the start and the end of a
    launch only.  On a (partition of a) RectangleMesh of triangles a block is a tile of 8 x 4 quads = 64 triangles, aligned with the
    LOCAL extent of the cells (a partition's ghost columns start a tile of their own instead of cutting through the global tile
    grid), tiles along a Hilbert curve: 24 rim facets per block (31 at most next to a partition's cuts) where the 16 x 2 blocks of
    the device numbering have 36 (68) - a third fewer granules to publish and to poll, and four granule loads per lane and polling
    pass instead of eight or nine (profiles/r05s_flow_block_order.txt).  Other triangulations: ``bisection_block_order``;
    quadrilaterals (no flow kernel): ``auto_cell_order``."""
    b = mesh.cells.shape[0] if b is None else b
    k = np.asarray(mesh.cells).shape[1]
    if k != 3:
        return auto_cell_order(mesh, a, b)
    if a == 0 and b == mesh.cells.shape[0] and getattr(mesh, 'structured', False):
        g, nx, ny = np.arange(b, dtype=np.int64), mesh.nx, mesh.ny
    elif getattr(mesh, 'structured_parent', None) is not None:
        g = np.asarray(mesh.local_to_global, dtype=np.int64)[a:b]
        nx, ny = mesh.structured_parent
    else:
        # any other triangulation: recursive coordinate bisection of the centroids down to leaves of exactly 64 cells (the left
        # half of every cut takes a multiple of 64) - boxes of aspect <= 2, 27 rim facets per block on a Delaunay mesh (38 at most)
        # where 64 consecutive cells of the Hilbert curve have 33 (51)
        cen = np.asarray(mesh.vertex_xy)[np.asarray(mesh.cells)[a:b]].mean(axis=1)
        return bisection_block_order(cen)
    q = g//2
    i, j = q % nx, q//nx
    il, jl = i - i.min(), j - j.min()
    bx, by = 8, 4
    order = max(1, int(np.ceil(np.log2(max(il.max()//bx + 2, jl.max()//by + 2)))))
    d = hilbert_index(il//bx, jl//by, order)
    return np.lexsort((g, il % bx, jl % by, d))


def fused_tile_order(mesh):
    """The order the TILES of the fused stage pair (csrc/swe2d_fuse.h: 192 consecutive cells + their ring per workgroup) are cut
    from, over ALL cells of ``mesh``; ``None`` where the device numbering already is that order (a whole mesh).  A partition's
    numbering is [interior | send cells | ghost layer 1 | ghost layer 2 ...] (what the stage ranges need): cut from it, the strips
    along the cut and every ghost layer - one cell wide - would give tiles that are all ring.  Here every cell, owned or ghost,
    sits in its 16 x 6-quad tile of the parent RectangleMesh (the tiles of the owned interior's numbering: their lanes still read
    consecutive addresses), tiles along the Hilbert curve; any other partition: a Hilbert curve through the centroids of all cells."""
    if getattr(mesh, 'local_to_global', None) is None:
        return None
    k = np.asarray(mesh.cells).shape[1]
    if k != 3:
        return None
    parent = getattr(mesh, 'structured_parent', None)
    if parent is not None:
        return structured_subset_order(np.asarray(mesh.local_to_global), parent[0], parent[1], bx=16, by=6)
    cen = np.asarray(mesh.vertex_xy)[np.asarray(mesh.cells)].mean(axis=1)
    return hilbert_cell_order(cen)


def triple_tile_order(mesh, bx=11, by=8):
    """(order, tile starts) of the TWO-RING tiles of a RectangleMesh of triangles - or of a partition of one, over ALL its cells, owned and
    ghost - (csrc/swe2d_fuse.h, swe_fuse123_kernel: interior + facet neighbours + their facet neighbours in 256 lanes): patches of
    bx x by quads of the (parent) mesh along the Hilbert curve, row by row inside a patch.  11 x 8 quads = 176 triangles + rings of
    38 + 42 = 256 lanes exactly (measured against 12 x 7, 14 x 6, 16 x 5, 10 x 8 and against as many consecutive cells of the 16 x 6
    numbering as fit - 147 + 52 + 57, ragged - at 0.5 ... 4 M cells: profiles/r06l_triple_tiles.txt); a patch cut short by the mesh's
    edge or by a partition's is a tile of its own.  ``None`` for any other mesh."""
    if np.asarray(mesh.cells).shape[1] != 3:
        return None
    parent = getattr(mesh, 'structured_parent', None)
    if parent is not None and getattr(mesh, 'local_to_global', None) is not None:
        nx, ny = parent
        g = np.asarray(mesh.local_to_global, dtype=np.int64)
    elif getattr(mesh, 'structured', False):
        nx, ny = mesh.nx, mesh.ny
        g = np.arange(2*nx*ny, dtype=np.int64)
    else:
        return None
    ntx, nty = -(-nx//bx), -(-ny//by)
    lev = max(1, int(np.ceil(np.log2(max(ntx, nty) + 1))))
    ti, tj = np.meshgrid(np.arange(ntx), np.arange(nty), indexing='ij')
    d_tile = hilbert_index(ti.ravel(), tj.ravel(), lev)
    rank_tile = np.empty(ntx*nty, dtype=np.int64)
    rank_tile[np.argsort(d_tile, kind='stable')] = np.arange(ntx*nty)               # patches numbered along the curve
    rank_tile = rank_tile.reshape(ntx, nty)
    q = g//2
    i, j = q % nx, q//nx
    patch = rank_tile[i//bx, j//by]
    key = ((patch*by + j % by)*bx + i % bx)*2 + g % 2
    order = np.argsort(key, kind='stable')
    pk = patch[order]
    starts = np.nonzero(np.concatenate(([True], pk[1:] != pk[:-1])))[0]
    return order, starts


def first_touch_vertex_order(cells):
    """Vertex permutation: vertices numbered in the order the (re-ordered) cells first reference them.
    Returns ``vperm`` with new vertex i = old vertex vperm[i]."""
    flat = np.asarray(cells).reshape(-1)
    _, first = np.unique(flat, return_index=True)
    return flat[np.sort(first)]
