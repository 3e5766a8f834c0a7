"""
The reference's OWN example scripts, run with nothing changed but the import line (``from thetis import *`` ->
``from thetis_amd import *``): ``examples/channel2d/channel2d.py`` writes its fields as UFL expressions of
``SpatialCoordinate(mesh2d)`` and ``conditional`` (:36-58), which thetis_amd/expr.py evaluates as lazy numpy expressions
(VERDICT r04 "missing 8": with Python callables only, the surface resembled the reference's, it did not stay).

The scripts are READ from /root/reference at test time, never copied; the test skips where the reference is absent (the GPU box).
CPU: the arithmetic behind the device interface is the oracle's C restatement (tests/cpu_device.py through the device_cls seam).
"""
import os
import sys

import numpy as np
import pytest

REF = '/root/reference'


def _run_reference_script(path, monkeypatch, tmp_path, replace=()):
    from thetis_amd import solver2d
    from cpu_device import CpuSwe2dDevice
    src = open(path).read()
    assert 'from thetis import *' in src
    src = src.replace('from thetis import *', 'from thetis_amd import *')
    for a, b in replace:
        assert a in src, a
        src = src.replace(a, b)
    monkeypatch.setattr(solver2d.FlowSolver2d, '_device_cls', CpuSwe2dDevice, raising=False)
    monkeypatch.setenv('THETIS_REGRESSION_TEST', '1')
    monkeypatch.chdir(tmp_path)
    ns = {'__name__': '__main__', '__file__': path}
    exec(compile(src, path, 'exec'), ns)
    return ns


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'examples/channel2d/channel2d.py')), reason='the reference tree is not here')
def test_reference_channel2d_script_runs_with_the_import_line_changed(ref_so, monkeypatch, tmp_path):
    """examples/channel2d/channel2d.py in regression mode (t_end = 500 s): bathymetry and initial elevation from UFL-style
    expressions, automatic time step, volume check - the closed-channel known answer: volume conserved to round-off
    (test/barotropicChannel/test_closed_channel.py:77-78 bar 1e-12)."""
    ns = _run_reference_script(os.path.join(REF, 'examples/channel2d/channel2d.py'), monkeypatch, tmp_path,
                               replace=[("options.fields_to_export = ['uv_2d', 'elev_2d']", "options.fields_to_export = ['uv_2d', 'elev_2d']\noptions.no_exports = True")])
    s = ns['solver_obj']
    assert 500.0 - 1e-6 <= s.simulation_time < 520.0 and s.iteration > 50
    vol, rel = s.callbacks['export']['volume2d']()
    assert abs(rel) < 1e-12
    # the initial elevation: the ramp of the script's conditional(), CG-P1 interpolated and projected
    x = ns['mesh2d'].vertex_xy[:, 0]
    assert np.allclose(ns['elev_init'].dat.data_ro, np.where(x < 30e3, 6.0*(1 - x/30e3), 0.0))
    assert np.allclose(ns['bathymetry_2d'].dat.data_ro, 20.0 - 15.0*x/100e3)


def test_expressions_evaluate_like_numpy():
    from thetis_amd import (Constant, Function, RectangleMesh, SpatialCoordinate, as_vector, conditional, cos, exp, get_functionspace, pi,
                            sin, sqrt)
    mesh = RectangleMesh(6, 4, 3.0, 2.0)
    x, y = SpatialCoordinate(mesh)
    P1 = get_functionspace(mesh, 'CG', 1)
    X, Y = mesh.vertex_xy.T
    f = Function(P1).interpolate(2.0 + 3*x/Constant(3.0) - y**2 + sin(pi*x)*cos(y) + exp(-x)*sqrt(1 + y) - (-x)/2)
    assert np.allclose(f.dat.data_ro, 2.0 + X - Y**2 + np.sin(np.pi*X)*np.cos(Y) + np.exp(-X)*np.sqrt(1 + Y) + X/2)
    g = Function(P1).interpolate(conditional(x < 1.5, 1.0 - x, f*0.0 + 7.0))
    assert np.allclose(g.dat.data_ro, np.where(X < 1.5, 1.0 - X, 7.0))
    V = get_functionspace(mesh, 'DG', 1, vector=True)
    w = Function(V).interpolate(as_vector((0.5 - y, x - 0.5)))
    xy = V.node_xy()
    assert np.allclose(w.dat.data_ro, np.stack([0.5 - xy[:, 1], xy[:, 0] - 0.5], axis=1))
    with pytest.raises(TypeError):
        bool(x < 1.0)


def test_function_operands_are_evaluated_through_their_spaces():
    """A Function inside an expression (the reference's scripts: a CG bathymetry inside the initial elevation that is PROJECTED into
    P1DG, a P1DG field inside an expression interpolated into P1DG, a P0 field): evaluated at the requested points through its own
    space - CG -> DG injection, interpolation to the cells' quadrature points - not by array length."""
    from thetis_amd import Function, RectangleMesh, SpatialCoordinate, get_functionspace, sqrt
    mesh = RectangleMesh(6, 4, 3.0, 2.0)
    x, y = SpatialCoordinate(mesh)
    P1, P1DG, P0 = get_functionspace(mesh, 'CG', 1), get_functionspace(mesh, 'DG', 1), get_functionspace(mesh, 'DG', 0)
    bath = Function(P1).interpolate(10.0 - 2.0*x + y)                       # linear: P1 holds it exactly
    xy = P1DG.node_xy()
    # CG operand, DG target, nodal interpolation
    f = Function(P1DG).interpolate(bath*2.0 + x)
    assert np.allclose(f.dat.data_ro, 2.0*(10.0 - 2.0*xy[:, 0] + xy[:, 1]) + xy[:, 0])
    # CG operand, DG target, L2 PROJECTION: evaluated at the quadrature points (one per cell and call) - exact for a linear field
    g = Function(P1DG).project(0.5*bath - y)
    assert np.allclose(g.dat.data_ro, 0.5*(10.0 - 2.0*xy[:, 0] + xy[:, 1]) - xy[:, 1])
    # DG operand in a projection, a nonlinear expression of it: against numpy quadrature of the same thing
    h = Function(P1DG).project(sqrt(f + 1.0))
    from thetis_amd.function import triangle_quadrature
    bary, w = triangle_quadrature()
    fc = f.dat.data_ro.reshape(-1, 3)
    b = np.zeros_like(fc)
    for l, wq in zip(bary, w):
        b += wq*np.sqrt(fc @ l + 1.0)[:, None]*l[None, :]
    assert np.allclose(h.dat.data_ro.reshape(-1, 3), 3.0*(4.0*b - b.sum(axis=1, keepdims=True)))
    # a P0 field has as many values as ... nothing else here: injected cell-wise, and into a projection as the cell's constant
    c = Function(P0)
    c.dat.data[:] = np.arange(mesh.num_cells)
    k = Function(P1DG).interpolate(c + 0.0*x)
    assert np.array_equal(k.dat.data_ro, np.repeat(np.arange(mesh.num_cells, dtype=float), 3))
    assert np.allclose(Function(P1DG).project(c*1.0).dat.data_ro, np.repeat(np.arange(mesh.num_cells, dtype=float), 3))
    # CG target of a projection with a DG operand: the global P1 projection of a continuous linear field returns it
    assert np.allclose(Function(P1).project(f*1.0 - x).dat.data_ro, 2.0*bath.dat.data_ro)
    # a DG-P1 field cannot be read at CG nodes (it is two-valued there): refused, not guessed
    with pytest.raises(NotImplementedError):
        Function(P1).interpolate(f*1.0)
    # another mesh: refused
    other = RectangleMesh(6, 4, 3.0, 2.0)
    with pytest.raises(ValueError):
        Function(get_functionspace(other, 'DG', 1)).interpolate(bath*1.0)
    # outside interpolate / project there are no points to evaluate a Function at
    with pytest.raises(ValueError):
        (bath*1.0)(np.zeros(3), np.zeros(3))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'demos/demo_2d_tracer.py')), reason='the reference tree is not here')
def test_reference_demo_2d_tracer_script_runs_with_the_import_line_changed(ref_so, monkeypatch, tmp_path):
    """demos/demo_2d_tracer.py (BASELINE cfg 4's source: 40 x 40 quadrilaterals, tracer only, SSPRK33, LeVeque's bell + cone + slotted
    cylinder written with pow / min_value / conditional / And): everything up to the end of its time loop runs as it stands; its last
    four lines assemble UFL forms (`assemble(... * dx)`: not part of this build's surface) and are cut - the same relative L2 error
    is formed here from the nodal fields.  A quarter revolution (the script's t_end replaced) keeps the CPU run short."""
    path = os.path.join(REF, 'demos/demo_2d_tracer.py')
    tail = "q = solver_obj.fields.tracer_2d"
    src = open(path).read()
    assert tail in src
    cut = src[src.index(tail):]
    ns = _run_reference_script(path, monkeypatch, tmp_path, replace=[(cut, ''), ('t_end = 2*pi', 't_end = pi/2'),
                                                                     ("options.fields_to_export = labels", "options.fields_to_export = labels\noptions.no_exports = True")])
    s = ns['solver_obj']
    q, q0 = s.fields.tracer_2d, ns['q_init']
    assert abs(s.simulation_time - (np.pi/2 - np.pi/300.0)) < 2*np.pi/300.0 and s.iteration >= 149
    qd = q.dat.data_ro
    assert np.isfinite(qd).all() and 0.9 < qd.min() and qd.max() < 2.2
    # the shapes have turned by a quarter revolution about the centre: compare with the initial field evaluated at the back-rotated nodes
    xy = q.function_space().node_xy()
    xb, yb = 0.5 + (xy[:, 1] - 0.5), 0.5 - (xy[:, 0] - 0.5)
    from thetis_amd import And, conditional, cos, min_value, pi, sqrt
    from thetis_amd.expr import Expr
    X, Y = Expr(lambda x, y: x), Expr(lambda x, y: y)
    bell = 0.25*(1 + cos(pi*min_value(sqrt(pow(X - 0.25, 2) + pow(Y - 0.5, 2))/0.15, 1.0)))
    cone = 1.0 - min_value(sqrt(pow(X - 0.5, 2) + pow(Y - 0.25, 2))/0.15, 1.0)
    cyl = conditional(sqrt(pow(X - 0.5, 2) + pow(Y - 0.75, 2)) < 0.15, conditional(And(And(X > 0.475, X < 0.525), Y < 0.85), 0.0, 1.0), 0.0)
    exact = (1.0 + bell + cone + cyl)(xb, yb)
    err = np.sqrt(np.mean((qd - exact)**2))/np.sqrt(np.mean(exact**2))
    assert err < 0.12, err
    assert q0.dat.data_ro.shape[0] == 41*41
