"""
Mesh files: Gmsh MSH 2.2 ASCII, the format of every ``.msh`` the reference ships (demos/north_sea.msh,
examples/*/mesh*.msh; Firedrake's ``Mesh('file.msh')`` reads them through DMPlex [FD-assumed]).

Triangles (element type 2) and quadrilaterals (type 3) become cells; line elements (type 1) carry the physical tags that
become boundary markers (``ds(tag)`` in the reference).  Exterior facets without a line element get ``default_marker``.
"""
import numpy as np

from .mesh import Mesh2d

__all__ = ['read_gmsh', 'write_gmsh']

_NODES = {1: 2, 2: 3, 3: 4, 15: 1}          # element type -> number of nodes (the types a 2D mesh file contains)


def read_gmsh(path, default_marker=None, name=None):
    with open(path) as f:
        lines = f.read().split('\n')
    pos = {l.strip(): i for i, l in enumerate(lines) if l.startswith('$')}
    if '$MeshFormat' not in pos or '$Nodes' not in pos or '$Elements' not in pos:
        raise ValueError('{:}: not a Gmsh MSH file'.format(path))
    version = lines[pos['$MeshFormat'] + 1].split()
    if not version[0].startswith('2') or version[1] != '0':
        raise NotImplementedError('{:}: only MSH 2.x ASCII is supported (file says {:})'.format(path, ' '.join(version)))
    i = pos['$Nodes'] + 1
    n_nodes = int(lines[i])
    node_tab = np.array([l.split() for l in lines[i + 1:i + 1 + n_nodes]], dtype=np.float64)
    node_ids = node_tab[:, 0].astype(np.int64)
    xy = node_tab[:, 1:3]
    id2idx = np.full(node_ids.max() + 1, -1, dtype=np.int64)
    id2idx[node_ids] = np.arange(n_nodes)
    i = pos['$Elements'] + 1
    n_el = int(lines[i])
    cells, edges, edge_tags = [], [], []
    cell_type = None
    for l in lines[i + 1:i + 1 + n_el]:
        t = l.split()
        etype, ntags = int(t[1]), int(t[2])
        if etype not in _NODES:
            raise NotImplementedError('{:}: element type {:d} is not supported'.format(path, etype))
        nodes = [int(v) for v in t[3 + ntags:3 + ntags + _NODES[etype]]]
        if etype in (2, 3):
            if cell_type not in (None, etype):
                raise NotImplementedError('{:}: mixed triangle/quadrilateral meshes are not supported'.format(path))
            cell_type = etype
            cells.append(nodes)
        elif etype == 1:
            edges.append(nodes)
            edge_tags.append(int(t[3]) if ntags > 0 else 0)       # first tag = physical entity
    if not cells:
        raise ValueError('{:}: no triangles or quadrilaterals'.format(path))
    cells = id2idx[np.array(cells, dtype=np.int64)]
    used = np.unique(cells)                                       # drop nodes that no cell references
    remap = np.full(n_nodes, -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    tag_of = {}
    for (a, b), tag in zip(edges, edge_tags):
        a, b = remap[id2idx[a]], remap[id2idx[b]]
        tag_of[(min(a, b), max(a, b))] = tag
    mesh = Mesh2d(xy[used], remap[cells], name=name or path)
    # exterior facet markers from the line elements
    k = mesh.nodes_per_cell
    c, f = np.nonzero(mesh.cell_nbr < 0)
    va, vb = mesh.cells[c, f], mesh.cells[c, (f + 1) % k]
    for ci, fi, a, b in zip(c, f, va, vb):
        tag = tag_of.get((min(a, b), max(a, b)))
        if tag is None or tag <= 0:
            if default_marker is None:
                raise ValueError('{:}: exterior facet without a physical tag (pass default_marker)'.format(path))
            tag = default_marker
        mesh.cell_nbr[ci, fi] = -tag
    mesh.boundary_len = mesh._boundary_length()
    return mesh


def write_gmsh(mesh, path):
    """MSH 2.2 ASCII with one line element per exterior facet (physical tag = boundary marker)."""
    k = mesh.nodes_per_cell
    c, f = np.nonzero(mesh.cell_nbr < 0)
    with open(path, 'w') as out:
        out.write('$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n{:d}\n'.format(mesh.num_vertices))
        for i, (x, y) in enumerate(mesh.vertex_xy):
            out.write('{:d} {!r} {!r} 0\n'.format(i + 1, float(x), float(y)))
        out.write('$EndNodes\n$Elements\n{:d}\n'.format(len(c) + mesh.num_cells))
        e = 1
        for ci, fi in zip(c, f):
            a, b = mesh.cells[ci, fi] + 1, mesh.cells[ci, (fi + 1) % k] + 1
            tag = -mesh.cell_nbr[ci, fi]
            out.write('{:d} 1 2 {:d} {:d} {:d} {:d}\n'.format(e, tag, tag, a, b))
            e += 1
        etype = 2 if k == 3 else 3
        for cell in mesh.cells:
            out.write('{:d} {:d} 2 1 1 '.format(e, etype) + ' '.join(str(v + 1) for v in cell) + '\n')
            e += 1
        out.write('$EndElements\n')
