"""
Vertex-based P1DG slope limiter (thetis/limiter.py:48-198, Kuzmin 2010), device resident.

Inside the coupled time integrator the limiter runs on the tracer that already lives in HBM
(``swe2d_tracer_limit``: cell means -> per-vertex min/max by CSR gather incl. Thetis's boundary-facet means ->
per-cell scaling).  ``apply(field)`` on a stand-alone host ``Function`` (test/slopelimiter/test_slopelimiter.py usage)
round-trips through a small private device handle.
"""
import numpy as np

from .device import Swe2dDevice

__all__ = ['VertexBasedP1DGLimiter']


class VertexBasedP1DGLimiter(object):
    def __init__(self, p1dg_space, time_dependent_mesh=True, device_id=0):
        if p1dg_space.family != 'DG' or p1dg_space.degree != 1:
            raise AssertionError('function space must be one of [\'Discontinuous Lagrange\', \'DQ\'] of degree 1')
        self.P1DG = p1dg_space
        self.is_vector = p1dg_space.vector
        self.mesh = p1dg_space.mesh()
        self.device_id = device_id
        self._dev = None
        self._tid = None

    def _device(self):
        if self._dev is None:
            self._dev = Swe2dDevice(self.mesh, np.ones(self.mesh.num_vertices), 1.0, device_id=self.device_id)
            self._tid = self._dev.add_tracer()
        return self._dev

    def apply(self, field):
        """Applies the limiter on the given field (in place)"""
        dev = self._device()
        data = field.dat.data
        comps = [data[:, i] for i in range(data.shape[1])] if self.is_vector else [data]
        for comp in comps:
            dev.tracer_set_state(self._tid, comp.reshape(-1, dev.npc))
            dev.tracer_limit(self._tid)
            comp[...] = dev.tracer_get_state(self._tid).reshape(comp.shape)
