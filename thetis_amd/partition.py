"""
Element partitioning for one-process-per-GPU runs.

The DG stencil couples facet neighbours only (every dS term of thetis/shallowwater_eq.py:363-366,424-427,480-488 uses
'+'/'-' traces), so a partition = owned cells + ONE layer of facet-adjacent ghost cells - the same overlap Firedrake's
DMPlex gives the reference [FD-assumed].  Local cell order on every rank:

    [ interior owned | boundary owned (touch a ghost) | ghosts grouped by owner rank ]

so the interior range can be computed while the halo is in flight.  Send lists and ghost blocks are both ordered by
global cell id, which makes the layout deterministic on both sides without any handshake: every rank builds its own
partition from the (replicated) global mesh.
"""
import numpy as np

__all__ = ['strip_owner', 'LocalPartition', 'build_partition']


def strip_owner(mesh, n_parts, axis=0):
    """Contiguous strips of (almost) equal cell count along ``axis`` (<= 2 peers per rank: one xGMI link each)."""
    c = mesh.cell_xy().mean(axis=1)
    # both triangles of a quad share the strip: sort by the quad-level coordinate first, cell id second
    order = np.lexsort((np.arange(mesh.num_cells), np.round(c[:, axis], 9)))
    owner = np.empty(mesh.num_cells, dtype=np.int32)
    bounds = np.linspace(0, mesh.num_cells, n_parts + 1).astype(np.int64)
    for p in range(n_parts):
        owner[order[bounds[p]:bounds[p + 1]]] = p
    return owner


class LocalPartition(object):
    """Mesh-like view of one rank's cells (owned + ghosts) with halo bookkeeping."""

    def __init__(self, rank, n_parts):
        self.rank, self.n_parts = rank, n_parts

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def num_vertices(self):
        return self.vertex_xy.shape[0]

    @property
    def n_ghost(self):
        return self.num_cells - self.n_owned

    def cell_xy(self):
        return self.vertex_xy[self.cells]


def build_partition(mesh, owner, rank):
    """Local partition of ``rank`` given the global ``owner`` array."""
    owner = np.asarray(owner)
    n_parts = int(owner.max()) + 1
    nbr = mesh.cell_nbr
    mine = np.nonzero(owner == rank)[0]
    nb = nbr[mine]                                            # (n_mine, k) global ids
    valid = nb >= 0
    nb_owner = np.where(valid, owner[np.where(valid, nb, 0)], rank)
    touches_ghost = (nb_owner != rank).any(axis=1)
    interior = mine[~touches_ghost]
    boundary = mine[touches_ghost]
    ghost_ids = np.unique(nb[valid & (nb_owner != rank)])
    ghost_owner = owner[ghost_ids]
    gorder = np.lexsort((ghost_ids, ghost_owner))             # by owner rank, then global id
    ghost_ids = ghost_ids[gorder]
    ghost_owner = ghost_owner[gorder]

    part = LocalPartition(rank, n_parts)
    local_global = np.concatenate([interior, boundary, ghost_ids]).astype(np.int64)
    part.local_to_global = local_global
    part.n_interior = len(interior)
    part.n_owned = len(mine)
    g2l = np.full(mesh.num_cells, -1, dtype=np.int64)
    g2l[local_global] = np.arange(len(local_global))

    # vertices actually used
    cells_g = mesh.cells[local_global]
    used = np.unique(cells_g)
    v_g2l = np.full(mesh.num_vertices, -1, dtype=np.int64)
    v_g2l[used] = np.arange(len(used))
    part.vertex_global = used
    part.vertex_xy = np.ascontiguousarray(mesh.vertex_xy[used])
    part.cells = np.ascontiguousarray(v_g2l[cells_g].astype(np.int32))

    nb_l = nbr[local_global].astype(np.int64)
    pos = nb_l >= 0
    mapped = np.where(pos, g2l[np.where(pos, nb_l, 0)], nb_l)
    # ghosts are never updated: neighbours of a ghost that are not local are irrelevant; point them at a wall
    mapped[(mapped < 0) & pos] = -1
    assert np.all(mapped[:part.n_owned][pos[:part.n_owned]] >= 0)
    part.cell_nbr = np.ascontiguousarray(mapped.astype(np.int32))
    part.cell_nbr_facet = np.ascontiguousarray(mesh.cell_nbr_facet[local_global])
    part.boundary_len = dict(mesh.boundary_len)
    part.boundary_markers = mesh.boundary_markers

    # ghost blocks by peer
    part.recv = {}
    off = 0
    for q in np.unique(ghost_owner):
        cnt = int((ghost_owner == q).sum())
        part.recv[int(q)] = (off, cnt)
        off += cnt
    # send lists: my cells that are ghosts of peer q, ordered by global id (the order q stores them in)
    part.send = {}
    send_cells = []
    off = 0
    for q in range(n_parts):
        if q == rank:
            continue
        theirs = np.nonzero(owner == q)[0]
        nbq = nbr[theirs]
        v = nbq >= 0
        hit = np.unique(nbq[v & (owner[np.where(v, nbq, 0)] == rank)])
        if len(hit):
            loc = g2l[hit]
            part.send[q] = (off, len(hit))
            send_cells.append(loc)
            off += len(hit)
    part.send_cells = (np.concatenate(send_cells) if send_cells else np.zeros(0, dtype=np.int64)).astype(np.int32)
    part.peers = sorted(set(part.send) | set(part.recv))
    return part
