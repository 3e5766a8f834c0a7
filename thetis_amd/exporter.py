"""
Field export and restart files around the hot path (thetis/exporter.py, thetis/solver2d.py:704-730,820-921).

* ``VTKExporter``: ParaView ``.vtu`` (XML header + raw appended binary data) per export + a ``.pvd`` collection, directory/file naming of the
  reference (``<outputdir>/<Filename>/<Filename>_<ix>.vtu``, exporter.py:64-120).  DG fields are written cell by cell
  (duplicated points), i.e. exactly the discontinuous data.
* Partitioned runs (one process per GPU, thetis_amd/spmd.py): reading a field for export gathers the owned cells of every rank
  (collective), rank 0 writes - the files are those of the single-device run; the other ranks only count the export index.
* ``CheckpointExporter``: the reference stores restart files with Firedrake's ``CheckpointFile`` (HDF5, exporter.py:123-242);
  neither h5py nor Firedrake exists here, so the same information (nodal data in the C-ABI host layout, export index,
  simulation time) goes to ``<outputdir>/hdf5/<Filename>_<ix:05d>.npz``.  NOT interchangeable with the reference's files.
"""
import os
from collections import OrderedDict

import numpy as np

__all__ = ['field_metadata', 'VTKExporter', 'CheckpointExporter', 'ExportManager']

# thetis/field_defs.py:37-42,67-72 (+ tracer metadata comes from options.add_tracer_2d)
field_metadata = {
    'uv_2d': {'name': 'Depth averaged velocity', 'shortname': 'Depth averaged velocity', 'unit': 'm s-1', 'filename': 'Velocity2d'},
    'elev_2d': {'name': 'Water elevation', 'shortname': 'Elevation', 'unit': 'm', 'filename': 'Elevation2d'},
}


class VTKExporter(object):
    def __init__(self, func_name, outputdir, filename, next_export_ix=0, writes=True):
        self.func_name, self.filename = func_name, filename
        self.dir = os.path.join(outputdir, filename)
        self.next_export_ix = next_export_ix
        self.entries = []
        if writes:
            os.makedirs(self.dir, exist_ok=True)

    def set_next_export_ix(self, ix):
        self.next_export_ix = ix

    def export(self, function, time=None):
        fs = function.function_space()
        mesh = fs.mesh()
        vals = function.cell_node_values()                     # (N, k[, 2])
        n, k = vals.shape[0], vals.shape[1]
        pts = np.concatenate([mesh.cell_xy().reshape(-1, 2), np.zeros((n*k, 1))], axis=1)
        fname = '{:s}_{:d}.vtu'.format(self.filename, self.next_export_ix)
        if vals.ndim == 3:
            data = np.concatenate([vals.reshape(-1, 2), np.zeros((n*k, 1))], axis=1)
            point_data = ('<PointData Vectors="{0}">\n<DataArray type="Float64" Name="{0}" NumberOfComponents="3" '
                          'format="appended" offset="{1}"/>\n')
        else:
            data = vals.reshape(-1, 1)
            point_data = '<PointData Scalars="{0}">\n<DataArray type="Float64" Name="{0}" format="appended" offset="{1}"/>\n'
        # raw appended binary blocks (UInt64 byte count + little-endian data): ParaView / VTK read them natively and a
        # million-cell export takes a fraction of a second instead of the tens of seconds of ASCII
        blocks = [np.ascontiguousarray(pts, dtype='<f8'),
                  np.ascontiguousarray(np.arange(n*k), dtype='<i4'),
                  np.ascontiguousarray((np.arange(n) + 1)*k, dtype='<i4'),
                  np.full(n, 5 if k == 3 else 9, dtype='u1'),
                  np.ascontiguousarray(data, dtype='<f8')]
        offsets, off = [], 0
        for blk in blocks:
            offsets.append(off)
            off += 8 + blk.nbytes
        with open(os.path.join(self.dir, fname), 'wb') as f:
            w = lambda text: f.write(text.encode('ascii'))
            w('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" '
              'header_type="UInt64">\n')
            w('<UnstructuredGrid>\n<Piece NumberOfPoints="{:d}" NumberOfCells="{:d}">\n'.format(n*k, n))
            w('<Points>\n<DataArray type="Float64" NumberOfComponents="3" format="appended" offset="{:d}"/>\n</Points>\n'.format(offsets[0]))
            w('<Cells>\n<DataArray type="Int32" Name="connectivity" format="appended" offset="{:d}"/>\n'.format(offsets[1]))
            w('<DataArray type="Int32" Name="offsets" format="appended" offset="{:d}"/>\n'.format(offsets[2]))
            w('<DataArray type="UInt8" Name="types" format="appended" offset="{:d}"/>\n</Cells>\n'.format(offsets[3]))
            w(point_data.format(self.func_name, offsets[4]))
            w('</PointData>\n</Piece>\n</UnstructuredGrid>\n<AppendedData encoding="raw">\n_')
            for blk in blocks:
                f.write(np.array([blk.nbytes], dtype='<u8').tobytes())
                f.write(blk.tobytes())
            w('\n</AppendedData>\n</VTKFile>\n')
        self.entries.append((self.next_export_ix if time is None else time, fname))
        with open(os.path.join(self.dir, self.filename + '.pvd'), 'w') as f:
            f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n<Collection>\n')
            for t, name in self.entries:
                f.write('<DataSet timestep="{:}" part="0" file="{:s}"/>\n'.format(t, name))
            f.write('</Collection>\n</VTKFile>\n')
        self.next_export_ix += 1


class CheckpointExporter(object):
    """Stand-in for ``HDF5Exporter`` (exporter.py:123-242): one ``.npz`` per export index."""

    def __init__(self, outputdir, filename_prefix, next_export_ix=0, writes=True):
        self.dir, self.prefix = outputdir, filename_prefix
        self.next_export_ix = next_export_ix
        if writes:
            os.makedirs(self.dir, exist_ok=True)

    def set_next_export_ix(self, ix):
        self.next_export_ix = ix

    def gen_filename(self, iexport):
        return os.path.join(self.dir, '{:s}_{:05d}.npz'.format(self.prefix, iexport))

    def export_as_index(self, iexport, function, time=None):
        np.savez(self.gen_filename(iexport), data=function.dat.data_ro, time=np.nan if time is None else float(time),
                 cells=function.function_space().mesh().cells)

    def export(self, function, time=None):
        self.export_as_index(self.next_export_ix, function, time=time)
        self.next_export_ix += 1

    def load(self, iexport, function):
        path = self.gen_filename(iexport)
        if not os.path.exists(path):
            raise IOError('checkpoint {:s} does not exist'.format(path))
        d = np.load(path)
        if not np.array_equal(d['cells'], function.function_space().mesh().cells):
            raise ValueError('checkpoint {:s} was written for a different mesh'.format(path))
        function.assign(d['data'])
        t = float(d['time'])
        return {} if np.isnan(t) else {'time': t}


class ExportManager(object):
    """Helper object for exporting multiple fields simultaneously (exporter.py:245-386)."""

    def __init__(self, outputdir, fields_to_export, functions, metadata, export_type='vtk', next_export_ix=0, comm=None):
        self.exporters = OrderedDict()
        self.functions = dict(functions)
        self.comm = comm
        writes = comm is None or comm.rank == 0
        for key in fields_to_export:
            field = self.functions.get(key)
            if field is None or not hasattr(field, 'cell_node_values'):
                continue
            meta = metadata[key]
            if export_type == 'vtk':
                self.exporters[key] = VTKExporter(meta['shortname'], outputdir, meta['filename'], next_export_ix, writes=writes)
            else:
                self.exporters[key] = CheckpointExporter(outputdir, meta['filename'], next_export_ix, writes=writes)

    def set_next_export_ix(self, ix):
        for e in self.exporters.values():
            e.set_next_export_ix(ix)

    def export(self, time=None):
        writes = self.comm is None or self.comm.rank == 0
        for key, e in self.exporters.items():
            f = self.functions[key]
            f.dat.data_ro                   # refreshes a device-resident field; on partitioned runs a gather EVERY rank takes part in
            if writes:
                e.export(f, time=time)
            else:
                e.next_export_ix += 1
