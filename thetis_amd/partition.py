"""
Element partitioning for one-process-per-GPU runs.

The DG stencil couples facet neighbours only (every dS term of thetis/shallowwater_eq.py:363-366,424-427,480-488 uses
'+'/'-' traces).  Firedrake/PyOP2 exchange a one-cell halo before every par_loop, i.e. once per RK stage [FD-assumed].
On 8 MI355X a stage of the 1M-triangle channel is ~6 us of work, less than one RCCL call, so the exchange is made
ONCE PER TIME STEP instead: every rank keeps ``halo_depth`` = 3 layers of facet-adjacent ghost cells (one per SSPRK33
stage) and recomputes them redundantly - stage 1 updates owned + layers 1,2, stage 2 owned + layer 1, stage 3 the owned
cells only (cfg 3: 2 x 1500 extra cell-updates per rank and step, 0.5 %).  Local cell order on every rank:

    [ interior owned | send owned (in some peer's halo; by decreasing distance from the cut) | ghost layer 1 | layer 2 | layer 3 ]

so that stage 3 can update the send cells first, start the exchange, and update the interior while it is in flight.
Send lists and receive lists are both ordered by global cell id, which makes the layout deterministic on both sides
without any handshake: every rank builds its own partition from the (replicated) global mesh.
"""
import numpy as np

__all__ = ['strip_owner', 'rcb_owner', 'LocalPartition', 'build_partition']


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


def rcb_owner(mesh, n_parts):
    """Recursive coordinate bisection of the cell centroids (general meshes): the cell set is split along its
    longer extent into two parts whose sizes follow the split of ``n_parts`` (any count, not only powers of two), and so
    on.  Compact parts => short halos; a part then has up to 8 neighbours on a 2D mesh (strips: 2).  Ties are broken by
    cell id, so every rank computes the same owner array."""
    c = mesh.cell_xy().mean(axis=1)
    owner = np.zeros(mesh.num_cells, dtype=np.int32)

    def split(ids, first, count):
        if count == 1:
            owner[ids] = first
            return
        left = count//2
        ext = c[ids].max(axis=0) - c[ids].min(axis=0)
        axis = 0 if ext[0] >= ext[1] else 1
        order = ids[np.lexsort((ids, np.round(c[ids, axis], 9)))]
        n_left = int(round(len(ids)*left/float(count)))
        split(order[:n_left], first, left)
        split(order[n_left:], first + left, count - left)
    split(np.arange(mesh.num_cells), 0, int(n_parts))
    return owner


class LocalPartition(object):
    """Mesh-like view of one rank's cells (owned + ghost layers) with halo bookkeeping."""

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

    def stage_range(self, i_stage, depth=3):
        """Cells [0, end) that stage ``i_stage`` has to update so that the owned cells are right after stage 3, when the
        stage input is valid on ``depth`` ghost layers (3 = one per SSPRK33 stage; a deeper halo only serves the limiter)."""
        depth = min(depth, len(self.layer_sizes))
        keep = max(0, depth - 1 - i_stage)              # ghost layers still needed after this stage
        return self.n_owned + int(sum(self.layer_sizes[:keep]))

    def owned_prefix(self, min_dist):
        """Number of leading local cells that are owned and at facet (or vertex) distance >= ``min_dist`` from every
        non-owned cell; ``min_dist`` <= 1: all owned cells, beyond the halo depth: the interior cells."""
        d = int(min(max(min_dist, 0), len(self.owned_dist_ge) - 1))
        return int(self.owned_dist_ge[d])

    def reorder_ranges(self):
        """Boundaries a device-side renumbering must not cross: the distance bands of the owned cells."""
        return tuple(sorted(set(int(x) for x in self.owned_dist_ge)))

    def layer_end(self, n_layers):
        """owned cells + the first ``n_layers`` ghost layers"""
        return self.n_owned + int(sum(self.layer_sizes[:n_layers]))


class _VertexAdjacency(object):
    """cells around every (topological) vertex, CSR"""

    def __init__(self, topo_cells):
        n, k = topo_cells.shape
        flat = topo_cells.ravel()
        order = np.argsort(flat, kind='stable')
        self.cell = (order//k).astype(np.int64)
        self.off = np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=int(flat.max()) + 1))])
        self.topo_cells = topo_cells

    def neighbours(self, cells):
        """all cells sharing a vertex with any of ``cells``"""
        v = np.unique(self.topo_cells[cells])
        starts, ends = self.off[v], self.off[v + 1]
        idx = np.concatenate([np.arange(a, b) for a, b in zip(starts, ends)]) if len(v) else np.zeros(0, dtype=np.int64)
        return np.unique(self.cell[idx])


def _halo_layers(nbr, inside, depth, vadj=None):
    """Layers 1..depth around the cell set ``inside`` (boolean mask): facet distance, or vertex distance when a
    _VertexAdjacency is given (a superset: the vertex-based limiter needs every cell around a vertex)."""
    dist = np.where(inside, 0, -1).astype(np.int32)
    frontier = np.nonzero(inside)[0]
    layers = []
    for d in range(1, depth + 1):
        if vadj is not None:
            nb = vadj.neighbours(frontier)
        else:
            nb = nbr[frontier].ravel()
            nb = nb[nb >= 0]
        new = np.unique(nb[dist[nb] < 0])
        dist[new] = d
        layers.append(new)
        frontier = new
    return layers, dist


def build_partition(mesh, owner, rank, halo_depth=3, adjacency='facet'):
    """Local partition of ``rank`` given the global ``owner`` array.  ``adjacency='vertex'`` builds the ghost layers by
    vertex distance (coupled runs with the vertex-based limiter use halo_depth=4, adjacency='vertex')."""
    owner = np.asarray(owner)
    n_parts = int(owner.max()) + 1
    nbr = mesh.cell_nbr
    mine_mask = owner == rank
    vadj = None
    if adjacency == 'vertex':
        topo = getattr(mesh, 'topo_vertex', None)
        vadj = _VertexAdjacency(np.asarray(mesh.cells if topo is None else np.asarray(topo)[mesh.cells], dtype=np.int64))
    elif adjacency != 'facet':
        raise ValueError("adjacency must be 'facet' or 'vertex'")
    layers, _ = _halo_layers(nbr, mine_mask, halo_depth, vadj)
    # which of my cells do the peers need?  (cells within halo_depth of the peer's owned set)
    send_global = {}
    for q in range(n_parts):
        if q == rank:
            continue
        _, dist_q = _halo_layers(nbr, owner == q, halo_depth, vadj)
        hit = np.nonzero(mine_mask & (dist_q > 0))[0]
        if len(hit):
            send_global[q] = np.sort(hit)
    in_send = np.zeros(mesh.num_cells, dtype=bool)
    for g in send_global.values():
        in_send[g] = True
    mine = np.nonzero(mine_mask)[0]
    interior = mine[~in_send[mine]]
    send_owned = mine[in_send[mine]]
    # send cells by decreasing distance d from the nearest non-owned cell (d = halo_depth ... 1): the cells at distance
    # >= d are then a PREFIX of the local numbering, which is what overlapping the exchange with the first stages of the
    # next step needs (stage g of a step reads ghost data only through cells at distance <= g + 1)
    _, dist_out = _halo_layers(nbr, ~mine_mask, halo_depth, vadj)
    d_send = dist_out[send_owned]
    assert (d_send >= 1).all()
    send_owned = send_owned[np.lexsort((send_owned, -d_send))]
    d_sorted = dist_out[send_owned]

    part = LocalPartition(rank, n_parts)
    ghost_layers = []
    for lay in layers:
        o = owner[lay]
        ghost_layers.append(lay[np.lexsort((lay, o))])           # by owner rank, then global id
    part.layer_sizes = [len(l) for l in ghost_layers]
    local_global = np.concatenate([interior, send_owned] + ghost_layers).astype(np.int64)
    part.local_to_global = local_global
    if getattr(mesh, 'structured', False):       # a partition of a RectangleMesh keeps the parent's tile numbering on the device
        part.structured_parent = (int(mesh.nx), int(mesh.ny))
    part.n_interior = len(interior)
    part.n_owned = len(mine)
    # owned_dist_ge[d] = number of owned cells at distance >= d from the non-owned cells (a prefix), d = 0 .. halo_depth + 1
    part.owned_dist_ge = np.array([len(interior) + int((d_sorted >= d).sum()) for d in range(halo_depth + 2)], dtype=np.int64)
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
    # topological vertex ids (periodic meshes identify vertices) of the local vertices, compact local numbering
    topo = getattr(mesh, 'topo_vertex', None)
    if topo is not None:
        _, tinv = np.unique(np.asarray(topo)[used], return_inverse=True)
        part.topo_vertex = tinv.astype(np.int64)
    else:
        part.topo_vertex = np.arange(len(used), dtype=np.int64)
    part.halo_depth, part.adjacency = halo_depth, adjacency
    part.affine = bool(getattr(mesh, 'affine', True))        # of the GLOBAL mesh: every rank takes the same kernel family

    nb_l = nbr[local_global].astype(np.int64)
    pos = nb_l >= 0
    mapped = np.where(pos, g2l[np.where(pos, nb_l, 0)], nb_l)
    # only the outermost ghost layer has neighbours that are not local; it is never updated: point them at a wall
    outer = (mapped < 0) & pos
    n_inner = part.n_owned + int(sum(part.layer_sizes[:-1])) if halo_depth > 0 else part.n_owned
    assert not outer[:n_inner].any(), 'a cell that is updated has a non-local neighbour'
    mapped[outer] = -1
    part.cell_nbr = np.ascontiguousarray(mapped.astype(np.int32))
    part.cell_nbr_facet = np.ascontiguousarray(mesh.cell_nbr_facet[local_global])
    part.boundary_len = dict(mesh.boundary_len)
    part.boundary_markers = mesh.boundary_markers

    # send lists (segments of one buffer, by peer) and receive lists (local ghost ids in the order the peer sends them)
    part.send, part.recv = {}, {}
    send_cells, recv_cells = [], []
    off = 0
    for q in sorted(send_global):
        loc = g2l[send_global[q]]
        part.send[q] = (off, len(loc))
        send_cells.append(loc)
        off += len(loc)
    ghosts = np.concatenate(ghost_layers) if ghost_layers else np.zeros(0, dtype=np.int64)
    ghost_owner = owner[ghosts]
    off = 0
    for q in np.unique(ghost_owner):
        gq = np.sort(ghosts[ghost_owner == q])                    # global ids, the order peer q packs them in
        part.recv[int(q)] = (off, len(gq))
        recv_cells.append(g2l[gq])
        off += len(gq)
    part.send_cells = (np.concatenate(send_cells) if send_cells else np.zeros(0, dtype=np.int64)).astype(np.int32)
    part.recv_cells = (np.concatenate(recv_cells) if recv_cells else np.zeros(0, dtype=np.int64)).astype(np.int32)
    part.peers = sorted(set(part.send) | set(part.recv))
    return part
