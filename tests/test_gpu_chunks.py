"""GPU: one mesh stepped chunk by chunk on ONE handle (tools/chunkbench.py: every strip a cell range followed by copies of its
neighbours' three facet layers, stages on shrinking ranges, the copies refreshed by swe2d_halo_pack / _unpack with both lists on
the same handle - include/swe2d.h: on a handle without ghost cells the receive list may name any cell) gives the bits of plain
stepping.  The measurement this served is DESIGN_ANNEX.md A5 (chunked stepping: slower); the entry-point semantics stay pinned here."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('chunks', [2, 5])
def test_chunked_stepping_gives_the_bits_of_plain_stepping(hip_lib, monkeypatch, chunks):
    import torch
    import chunkbench
    from helpers import channel_case
    from thetis_amd.device import Swe2dDevice
    monkeypatch.setenv('THETIS_AMD_FLOW', '0')
    mesh, bath, uv, eta = channel_case(nx=120, ny=40, seed=9)
    cm = chunkbench.build(mesh, chunks)
    assert cm.num_cells > mesh.num_cells and len(cm.send) == len(cm.recv) == cm.num_cells - mesh.num_cells
    dev = Swe2dDevice(cm, np.asarray(bath)[cm.vertex_global], 0.05, boundary_len=cm.boundary_len, ranges=cm.ranges)
    dev.set_state(uv[cm.local_to_global], eta[cm.local_to_global])
    dev.halo_setup(cm.send, cm.recv)
    buf = torch.empty(len(cm.recv)*9, dtype=torch.float64, device='cuda')
    n_steps = 6
    for _ in range(n_steps):
        for chunk in cm.stages:
            for s, (a, b) in enumerate(chunk):
                dev.solve_stage_cells(s, a, b)
        dev.halo_pack(0, buf.data_ptr())
        dev.halo_unpack(0, buf.data_ptr())
    dev.synchronize()
    cu, ce = dev.get_state()
    dev.close()
    plain = Swe2dDevice(mesh, bath, 0.05, boundary_len=mesh.boundary_len)
    plain.set_state(uv, eta)
    plain.advance(n_steps)
    pu, pe = plain.get_state()
    plain.close()
    g = cm.local_to_global[cm.owned]
    assert np.isfinite(pu).all() and np.abs(pu).max() > 0
    assert np.array_equal(cu[cm.owned], pu[g]) and np.array_equal(ce[cm.owned], pe[g])
