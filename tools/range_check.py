#!/usr/bin/env python
"""Range-check pass over the kernels (tools/range_check.sh): exercises every kernel family of the library on small meshes
against the -DSWE_RANGE_CHECK build, in which every raw-buffer access of the stage / tracer / viscosity / diagnostics kernels
is tested against the table of the library's own allocations (swe2d_kernels.h).  Prints the report and exits non-zero on a
violation.  THETIS_AMD_RANGE_SELFTEST=1 (negative control): the table records HALF of every allocation - violations expected.
(The image has no AddressSanitizer-enabled HIP runtime: an ASAN build of the library compiles for gfx950:xnack+ but cannot
be loaded, so this is the sanitizer pass of SURVEY.md section 5 for the device code.)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def exercise():
    from helpers import channel_case, delaunay_case, quad_case
    from thetis_amd import _lib, ordering
    from thetis_amd.device import Swe2dDevice
    n_launch = 0
    for name, case in (('triangles', channel_case(nx=23, ny=11, seed=1)), ('quadrilaterals', quad_case(nx=17, ny=9, seed=2)),
                       ('unstructured', delaunay_case(n_points=500, seed=3)[:4])):
        mesh, bath, uv, eta = case
        k = mesh.cells.shape[1]
        cxy = mesh.cell_xy()
        for variant in ('plain', 'open+fields', 'sources', 'viscosity', 'wetting-drying', 'tracers'):
            dev = Swe2dDevice(mesh, bath if variant != 'wetting-drying' else bath - 0.6*bath.max(), 0.05,
                              boundary_len=mesh.boundary_len)
            markers = mesh.boundary_markers
            if variant == 'open+fields':
                dev.set_bc(markers[0], {'elev': 0.2*np.sin(cxy[:, :, 1]/3e3)})
                dev.set_bc(markers[-1], {'un': 0.05, 'drag': 0.01})
            if variant == 'sources':
                dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
                dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones((mesh.num_cells, k)))
                dev.set_field(_lib.FIELD_WIND_STRESS, 0.1*np.ones((mesh.num_cells, k, 2)))
            if variant == 'viscosity':
                dev.set_viscosity(20.0 + 5.0*np.arange(mesh.num_vertices)/mesh.num_vertices, use_grad_div_viscosity_term=True)
                dev.set_bc(markers[0], {'un': 0.1})
            if variant == 'wetting-drying':
                dev.set_wetting_and_drying(0.5)
                dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
            dev.set_state(0.1*uv, 0.1*np.abs(eta))
            if variant == 'tracers':
                tid = dev.add_tracer()
                dev.tracer_set_state(tid, 1.0 + 0.1*np.random.default_rng(0).normal(size=(mesh.num_cells, k)))
                dev.tracer_set_diffusivity(tid, 5.0)
                dev.tracer_set_bc(tid, markers[0], 1.5)
                dev.tracer_set_bc_velocity(tid, markers[0], un=0.1*np.ones((mesh.num_cells, k)))
                dev.advance_coupled(2)
                dev.tracer_diagnostics(tid)
                n_launch += 20
            dev.advance(2)
            dev.advance_forward_euler(1)
            for i in range(3):                      # sub-range launches with ragged ends
                dev.solve_stage_cells(i, 0, mesh.num_cells//3 + 1)
                dev.solve_stage_cells(i, mesh.num_cells//3 + 1, mesh.num_cells)
            if dev.flow_supported():                # the dataflow launch (plane accesses are checked; granules use a bounded resource)
                dev.solve_flow([mesh.num_cells]*6)
                n_launch += 1
            if variant in ('plain', 'open+fields', 'sources'):
                # round 6: the fused stage kernels (csrc/swe2d_fuse.h: the tile tables are host-built indices into LDS and memory) -
                # stages 1 + 2 in one launch on triangles and quadrilaterals, all three on triangles, forced on these small meshes
                dev.set_option(_lib.OPT_FLOW, 0)
                for mode in ((1, 3, 33) if k == 3 else (1,)):
                    if mode == 33:                   # the two-ring tiles as patches (structured meshes: 5 x 3 quads; else bisection leaves of 40 cells)
                        mode = 3
                        tiles = ordering.triple_tile_order(mesh, 5, 3)
                        if tiles is None:
                            cen = mesh.cell_xy().mean(axis=1)
                            tiles = (ordering.bisection_block_order(cen, block=40), np.arange(0, mesh.num_cells, 40))
                        dev.fused_set_triple_tiles(*tiles)
                    dev.set_option(_lib.OPT_FUSED_STAGES, mode)
                    assert dev.fused_pair_info()[0] and (mode != 3 or dev.fused_triple_info()[0]), (name, variant, mode)
                    dev.advance(3)
                    n_launch += 6
                dev.set_option(_lib.OPT_FUSED_STAGES, None)
                dev.set_option(_lib.OPT_FLOW, None)
            dev.tendency()
            d = dev.diagnostics()
            assert np.isfinite(d).all(), (name, variant, d)
            dev.get_state()
            dev.close()
            n_launch += 30
            print('ok', name, variant, flush=True)
    # partitions: halo cells, owned / interior sub-ranges, pack / unpack
    from thetis_amd.partition import build_partition, rcb_owner
    mesh, bath, uv, eta = channel_case(nx=23, ny=11, seed=4)
    owner = rcb_owner(mesh, 3)
    for rank in range(3):
        p = build_partition(mesh, owner, rank)
        dev = Swe2dDevice(p, np.asarray(bath)[p.vertex_global], 0.05, n_owned=p.n_owned, boundary_len=p.boundary_len,
                          ranges=p.reorder_ranges())
        dev.halo_setup(p.send_cells, p.recv_cells)
        dev.set_state(0.1*uv[p.local_to_global], 0.1*eta[p.local_to_global])
        for i in range(3):
            inner = p.owned_prefix(3)
            dev.solve_stage_cells(i, 0, inner)
            dev.solve_stage_cells(i, inner, p.stage_range(i))
        # the fused stage pair on the partition's ranges (tiles cut from an order that mixes owned and ghost cells)
        from thetis_amd import ordering
        dev.set_option(_lib.OPT_FUSED_STAGES, 1)
        dev.fused_set_order(ordering.fused_tile_order(p))
        assert dev.fused_pair_info()[0]
        dev.solve_stage_pair_cells(p.stage_range(0), p.stage_range(1))
        dev.solve_stage_cells(2, 0, p.n_owned)
        # ... and whole steps in one launch each on the partition's two-ring tiles, cut as patches of the parent mesh (5 x 3 quads here),
        # the state buffers changing places (swe2d_solve_step_cells)
        dev.set_option(_lib.OPT_FUSED_STAGES, 3)
        tiles = ordering.triple_tile_order(p, 5, 3)
        if tiles is not None:
            dev.fused_set_triple_tiles(*tiles)
        assert dev.fused_step_info()[0]
        dev.solve_step_cells(p.n_owned)
        dev.diagnostics()
        dev.close()
        n_launch += 10
    print('ok partitions', flush=True)
    return n_launch


def main():
    from thetis_amd import _lib
    selftest = os.environ.get('THETIS_AMD_RANGE_SELFTEST') == '1'
    n_launch = 0
    try:
        n_launch = exercise()
    except Exception as e:                       # the negative control suppresses accesses: non-finite states are expected there
        if not selftest:
            raise
        print('negative control stopped at:', e)
    out = (ctypes.c_ulonglong*5)()
    lib = _lib.load()
    if not hasattr(lib, 'swe2d_debug_range_report'):
        raise SystemExit('not a -DSWE_RANGE_CHECK build: ' + _lib.LIB_PATH)
    lib.swe2d_debug_range_report.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert lib.swe2d_debug_range_report(out) == 0
    print('range check: ~{:d} launches, {:d} checked launches, {:d} violations{}'.format(
        n_launch, out[3], out[0], '' if not out[0] else ' (first: address 0x{:x}, swe2d_kernels.h/swe2d_sipg.h line {:d})'.format(out[1], out[2])))
    if selftest:
        sys.exit(0 if out[0] > 0 else 'negative control found no violation')
    sys.exit(1 if out[0] else 0)


if __name__ == '__main__':
    main()
