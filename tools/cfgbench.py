#!/usr/bin/env python
"""Per-configuration timings of the other BASELINE configs (not the headline bench line): cfg 1(ii)/quads, cfg 4
(coupled SWE + tracer + limiter), cfg 5 (wetting-drying), each with algorithmic bytes per step and the HBM fraction.
   python tools/cfgbench.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from thetis_amd import _lib                      # noqa: E402
from thetis_amd.device import Swe2dDevice        # noqa: E402
from thetis_amd.mesh import RectangleMesh        # noqa: E402


def timed(dev, fn, steps, prewarm=0.4):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm:        # settle the clocks (see bench.py)
        fn(20)
        dev.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        fn(steps)
        dev.synchronize()
        best = min(best, (time.perf_counter() - t0)/steps)
    return best


def report(name, n_cells, bytes_per_cell_step, t_step, extra=None):
    out = {'config': name, 'n_cells': n_cells, 'us_per_step': 1e6*t_step,
           'algorithmic_bytes_per_cell_step': bytes_per_cell_step,
           'achieved_GBs': bytes_per_cell_step*n_cells/t_step/1e9,
           'frac_of_8TBs': bytes_per_cell_step*n_cells/t_step/8e12}
    out.update(extra or {})
    print(json.dumps(out))


def quads(rng):
    # ---- quads: 1M quadrilaterals (cfg 1(ii) cell type at bench size): 248 / 344 / 344 = 936 B per cell per step
    # CFGBENCH_QUAD_N: cells per side (1000 = the bench size, 936 B x 3 state buffers = 288 MB: beyond the 256 MB Infinity Cache;
    # 800 = 640 k cells, 184 MB: inside it, like the 1 M triangles of cfg 2)
    nside = int(os.environ.get('CFGBENCH_QUAD_N', '1000'))
    meshq = RectangleMesh(nside, nside, 100e3, 100e3, quadrilateral=True)
    nq = meshq.num_cells
    cq = meshq.cell_xy()
    etaq = 0.5*np.exp(-((cq[:, :, 0] - 50e3)**2 + (cq[:, :, 1] - 50e3)**2)/(5e3)**2)
    dev = Swe2dDevice(meshq, np.full(meshq.num_vertices, 20.0), 0.25)
    dev.set_state(1e-3*rng.uniform(-1, 1, size=(nq, 4, 2)), etaq)
    fz = {'fused_pair': bool(dev.fused_pair_info()[0])}
    report('quadrilaterals SWE (DQ-1)', nq, 936.0, timed(dev, dev.advance, 50), fz)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    report('quadrilaterals SWE + Manning drag', nq, 936.0, timed(dev, dev.advance, 50), fz)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, None)
    # cfg 4 on its own cell type (demos/demo_2d_tracer.py is a quadrilateral mesh): tracer per stage 32 r + 32 w (+32 T0)
    # + 64 velocity + 56 static = 184 / 216 / 216 B; limiter ~ 32 + 8 + 8 + 8 + 32 + 32 + 16 = 136 B
    tid = dev.add_tracer()
    dev.tracer_set_state(tid, np.where(cq[:, :, 0] < 40e3, 0.0, 30.0))
    report('cfg4 quadrilaterals SWE + tracer + limiter', nq, 936.0 + 616.0 + 136.0,
           timed(dev, lambda k: dev.advance_coupled(k, tracer_only=False, use_limiter=True), 50), fz)
    report('cfg4 quadrilaterals tracer only + limiter (demo_2d_tracer mode)', nq, 616.0 + 136.0,
           timed(dev, lambda k: dev.advance_coupled(k, tracer_only=True, use_limiter=True), 50))
    dev.close()
    # ---- the same SWE step on GENERAL quadrilaterals (vertices moved by up to 20 % of a cell width: Jacobian per Gauss point,
    #      4 x 4 mass solve per cell - the AFFINE = false kernels)
    from thetis_amd.mesh import Mesh2d
    xy = meshq.vertex_xy.copy()
    inner = (xy[:, 0] > 1.0) & (xy[:, 0] < 100e3 - 1.0) & (xy[:, 1] > 1.0) & (xy[:, 1] < 100e3 - 1.0)
    xy[inner] += 20.0*rng.uniform(-1, 1, size=(int(inner.sum()), 2))
    warped = Mesh2d(xy, meshq.cells, marker_fn=None)
    warped.cell_nbr = meshq.cell_nbr
    warped.boundary_len = warped._boundary_length()
    assert not warped.affine
    dev = Swe2dDevice(warped, np.full(warped.num_vertices, 20.0), 0.25, boundary_len=warped.boundary_len)
    dev.set_state(1e-3*rng.uniform(-1, 1, size=(nq, 4, 2)), etaq)
    report('general quadrilaterals SWE (bilinear map, 4x4 mass solve)', nq, 936.0, timed(dev, dev.advance, 50),
           {'fused_pair': bool(dev.fused_pair_info()[0])})
    dev.close()


def cfg5(steps=50, profile_only=False):
    # ---- cfg 5: wetting-drying variant on the Balzano geometry, 500k triangles (+12 B alpha per cell-stage via vertices ~ +6)
    mesh5 = RectangleMesh(707, 354, 13800.0, 7200.0)
    n5 = mesh5.num_cells
    dev = Swe2dDevice(mesh5, mesh5.vertex_xy[:, 0]/2760.0, 0.1)
    dev.set_wetting_and_drying(0.4)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    dev.set_bc(2, {'elev': -0.5})
    dev.set_state(np.zeros((n5, 3, 2)), np.zeros((n5, 3)))
    if profile_only:                                      # a few steps for rocprofv3 --pmc
        dev.advance(steps)
        dev.synchronize()
    else:
        report('cfg5 triangles SWE wetting-drying + Manning + open bc', n5, 684.0 + 18.0, timed(dev, dev.advance, steps))
    dev.close()


def cfg5_parts(steps=50, profile_only=False):
    """cfg 5 taken apart: the Balzano geometry with neither / one / both of wetting-drying and Manning friction.  The four stage
    kernels have different template arguments, so a PMC pass (CFGBENCH_ONLY=cfg5_parts_profile under tools/pmc.sh) lists their
    instruction counts side by side: what each option costs per wave."""
    mesh5 = RectangleMesh(707, 354, 13800.0, 7200.0)
    n5 = mesh5.num_cells
    for wd, manning in ((False, False), (True, False), (False, True), (True, True)):
        # without wetting-drying the beach must stay wet: a deeper, gently sloping bed of the same shape
        bath = mesh5.vertex_xy[:, 0]/2760.0 + (0.0 if wd else 3.0)
        dev = Swe2dDevice(mesh5, bath, 0.1)
        if wd:
            dev.set_wetting_and_drying(0.4)
        if manning:
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        dev.set_bc(2, {'elev': -0.5})
        dev.set_state(np.zeros((n5, 3, 2)), np.zeros((n5, 3)))
        if profile_only:
            dev.advance(steps)
            dev.synchronize()
        else:
            report('cfg5 geometry, wetting-drying {:}, Manning {:}'.format('on' if wd else 'off', 'on' if manning else 'off'),
                   n5, 684.0 + (18.0 if wd else 0.0), timed(dev, dev.advance, steps))
        dev.close()


def main():
    only = os.environ.get('CFGBENCH_ONLY', '')          # e.g. 'quads' for kernel A/B runs
    if only == 'cfg5_parts':
        return cfg5_parts()
    if only == 'cfg5_parts_profile':
        return cfg5_parts(12, True)
    rng = np.random.default_rng(1234)
    if only == 'quads':
        return quads(rng)
    if only == 'cfg5':
        return cfg5()
    if only == 'cfg5_profile':
        return cfg5(12, True)
    # ---- cfg 2 reference point: triangles, SWE only (684 B per cell per step)
    nx = int(os.environ.get('CFGBENCH_NX', '1000'))       # 2000: the same channel with 4 M triangles (beyond the Infinity Cache)
    mesh = RectangleMesh(nx, nx//2, 100e3, 50e3)
    n = mesh.num_cells
    bath = np.full(mesh.num_vertices, 20.0)
    cxy = mesh.cell_xy()
    eta = 0.5*np.exp(-((cxy[:, :, 0] - 50e3)**2 + (cxy[:, :, 1] - 25e3)**2)/(5e3)**2) + 1e-3*rng.uniform(-1, 1, size=(n, 3))
    uv = 1e-3*rng.uniform(-1, 1, size=(n, 3, 2))
    dev = Swe2dDevice(mesh, bath, 0.25*1000.0/nx)
    dev.set_state(uv, eta)
    report('cfg2 triangles SWE', n, 684.0, timed(dev, dev.advance, 50))
    # ---- the same with the source terms of a tidal case: Coriolis field (24 B per cell and stage), Manning friction, wind stress (48 B)
    dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*(1.0 + cxy[:, :, 1]/50e3))
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
    dev.set_field(_lib.FIELD_WIND_STRESS, np.stack([0.1*np.sin(cxy[:, :, 0]/2e4), 0.05*np.cos(cxy[:, :, 1]/1e4)], axis=2))
    report('cfg2 + Coriolis field + Manning + wind stress', n, 684.0 + 3*72.0, timed(dev, dev.advance, 50),
           {'fused_pair': bool(dev.fused_pair_info()[0])})
    dev.set_field(_lib.FIELD_CORIOLIS, None)
    dev.set_scalar(_lib.SCALAR_MANNING_DRAG, None)
    dev.set_field(_lib.FIELD_WIND_STRESS, None)
    # ---- cfg 4: coupled SWE + 1 tracer + limiter.  Tracer per stage: 24 r + 24 w (+24 T0) + 48 velocity + 36 static
    #      = 132 / 156 / 156 B; limiter once per step: 24 r (means) + 8 w + 8 r + vertex bounds ~16 + 24 r + 24 w + 12 idx ~ 116 B
    tid = dev.add_tracer()
    dev.tracer_set_state(tid, np.where(cxy[:, :, 0] < 40e3, 0.0, 30.0))
    fz = {'fused_pair': bool(dev.fused_pair_info()[0])}
    report('cfg4 triangles SWE + tracer + limiter', n, 684.0 + 444.0 + 116.0,
           timed(dev, lambda k: dev.advance_coupled(k, tracer_only=False, use_limiter=True), 50), fz)
    report('cfg4 tracer only + limiter', n, 444.0 + 116.0,
           timed(dev, lambda k: dev.advance_coupled(k, tracer_only=True, use_limiter=True), 50), fz)
    report('tracer only, no limiter', n, 444.0,
           timed(dev, lambda k: dev.advance_coupled(k, tracer_only=True, use_limiter=False), 50), fz)
    # ---- optional SIPG passes (swe2d_sipg.h): per stage the pass re-reads the rows (+ neighbour rows through L2) and
    #      read-modify-writes them: viscosity 48 r + 48 r/w + 24 eta + 36 static = 204 B, tracer diffusion 24 + 48 + 36 = 108 B
    dev.tracer_set_diffusivity(tid, 10.0)
    report('tracer only + SIPG diffusion (no limiter)', n, 444.0 + 3*108.0,
           timed(dev, lambda k: dev.advance_coupled(k, tracer_only=True, use_limiter=False), 50))
    dev.tracer_set_diffusivity(tid, None)
    dev.set_viscosity(10.0)
    report('cfg2 + SIPG viscosity', n, 684.0 + 3*204.0, timed(dev, dev.advance, 50))
    dev.set_viscosity(None)
    dev.close()
    if only == 'tracers':                         # the triangle rows only (kernel A/B runs)
        return
    cfg5()
    quads(rng)



if __name__ == '__main__':
    main()
