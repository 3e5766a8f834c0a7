"""GPU: the dataflow stage loop (csrc/swe2d_flow.h: many stages in one launch, blocks wait for their neighbours' stage
counters instead of a kernel boundary) gives the bits of the stage launches it stands for."""
import numpy as np
import pytest

from helpers import channel_case, delaunay_case

pytestmark = pytest.mark.gpu


def _device(mesh, bath, dt, **kw):
    from thetis_amd.device import Swe2dDevice
    return Swe2dDevice(mesh, bath, dt, boundary_len=mesh.boundary_len, **kw)


def _by_stage(dev, ends):
    for s, end in enumerate(ends):
        dev.solve_stage_cells(s % 3, 0, end)


def _configure(dev, mesh, case):
    from thetis_amd import _lib
    k = mesh.cells.shape[1]
    cxy = mesh.cell_xy()
    if case in ('channel_open', 'sources', 'sources_large'):
        m = mesh.boundary_markers
        dev.set_bc(m[0], {'elev': 0.2*np.sin(cxy[:, :, 1]/3e3)})
        dev.set_bc(m[-1], {'un': 0.05, 'drag': 0.01})
        if len(m) > 2:
            dev.set_bc(m[1], {'flux': 30.0})
    if case in ('sources', 'sources_large'):
        dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones((mesh.num_cells, k)))
        dev.set_field(_lib.FIELD_WIND_STRESS, 0.1*np.ones((mesh.num_cells, k, 2)))


@pytest.mark.parametrize('case', ['channel', 'channel_open', 'unstructured', 'linear', 'no_lf', 'sources', 'sources_large',
                                  'large', 'periodic', 'tiny'])
def test_flow_launch_gives_the_bits_of_the_stage_launches(hip_lib, case):
    if case == 'unstructured':
        mesh, bath, uv, eta = delaunay_case(n_points=3000, seed=5)[:4]
    elif case == 'periodic':
        from thetis_amd.mesh import PeriodicRectangleMesh
        mesh = PeriodicRectangleMesh(61, 29, 100e3, 30e3, direction='x')
        rng = np.random.default_rng(17)
        bath = 20.0 + 2.0*np.sin(mesh.vertex_xy[:, 1]/5000.0)
        uv, eta = 0.5*rng.normal(size=(mesh.num_cells, 3, 2)), 0.5*rng.normal(size=(mesh.num_cells, 3))
    else:
        nx, ny = {'sources_large': (250, 125), 'large': (300, 200), 'tiny': (3, 2)}.get(case, (67, 31))
        mesh, bath, uv, eta = channel_case(nx=nx, ny=ny, seed=11)
    kw = {}
    if case == 'linear':
        kw['use_nonlinear_equations'] = False
    if case == 'no_lf':
        kw['use_lax_friedrichs_velocity'] = False
    out = []
    n_steps = 5
    for flow in (False, True):
        dev = _device(mesh, bath, 0.05 if case != 'unstructured' else 0.02, **kw)
        _configure(dev, mesh, case)
        assert dev.flow_supported() == (1 if case.startswith('sources') else 2)
        dev.set_state(uv, eta)
        ends = [dev.n_cells]*(3*n_steps)
        if flow:
            dev.solve_flow(ends[:6])            # two launches: the stage counters carry over
            dev.solve_flow(ends[6:])
        else:
            _by_stage(dev, ends)
        out.append(dev.get_state())
        assert dev.flow_timeouts() == 0
        dev.close()
    assert np.isfinite(out[0][0]).all() and np.abs(out[0][0]).max() > 0
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('poll', ['auto', '8', '9'])
def test_flow_polling_width_follows_the_widest_block(hip_lib, monkeypatch, poll):
    """A flow order in which some block has more than 64 rim facets (a mesh whose width is no multiple of the 16-quad tiles: the
    partial tiles at the edge make such blocks) takes the kernel instance with nine granule loads per lane and polling trip - a
    block that needs a second trip per pass paces the whole launch.  Same bits with eight (two trips) and nine loads, on a mesh
    with and on one without wide blocks; THETIS_AMD_FLOW_POLL forces the choice."""
    if poll != 'auto':
        monkeypatch.setenv('THETIS_AMD_FLOW_POLL', poll)
    for nx, ny in ((67, 31), (128, 64)):            # widest block: 67 / 36 rim facets
        mesh, bath, uv, eta = channel_case(nx=nx, ny=ny, seed=3)
        out = []
        for flow in (False, True):
            dev = _device(mesh, bath, 0.05)
            dev.set_state(uv, eta)
            ends = [dev.n_cells]*9
            if flow:
                dev.solve_flow(ends)
            else:
                _by_stage(dev, ends)
            out.append(dev.get_state())
            assert dev.flow_timeouts() == 0
            dev.close()
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize('shape', ['shrinking', 'ragged'])
def test_flow_launch_on_shrinking_ranges(hip_lib, shape):
    """The ranges of an exchange cycle: stage s updates [0, end_s), ends non-increasing and not block-aligned, every cell of a
    stage's range has its neighbours inside the previous stage's range (natural numbering: a cell's neighbours lie within one
    mesh row of 2 nx cells, so ranges that lose at least a row per stage qualify).  Blocks retire when the ranges have passed
    them, partially covered blocks keep their outer lanes' values; state buffer 0 ends as the stage launches leave it."""
    nx = 120
    mesh, bath, uv, eta = channel_case(nx=nx, ny=60, seed=4)
    n, row = mesh.num_cells, 2*nx
    if shape == 'shrinking':
        ends = [n - row*s for s in range(12)]
    else:
        ends = [n, n, n, n - row, n - 2*row, n - 10*row, n - 11*row, n - 12*row, 30*row, 29*row, 5*row, 4*row]
    out = []
    for flow in (False, True):
        dev = _device(mesh, bath, 0.05, reorder=None)
        dev.set_state(uv, eta)
        if flow:
            dev.solve_flow(ends)
        else:
            _by_stage(dev, ends)
        out.append(dev.get_state())
        assert dev.flow_timeouts() == 0
        dev.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_results_do_not_depend_on_the_flow_order(hip_lib):
    """Blocks of a scrambled order of the cells (random permutations of windows of 128 cells: most facets cross block rims) and
    of a reversed order: the same bits as the default blocks - and as the stage launches.  An order without any locality (more
    rim facets per block than the kernel's staging area holds) is refused, not run."""
    mesh, bath, uv, eta = channel_case(nx=50, ny=21, seed=6)
    rng = np.random.default_rng(3)
    n = mesh.num_cells
    scrambled = np.concatenate([a + rng.permutation(min(128, n - a)) for a in range(0, n, 128)])
    dev = _device(mesh, bath, 0.05)
    dev.flow_set_order(rng.permutation(n))
    assert dev.flow_supported() == 0
    dev.flow_set_order(np.arange(n))
    assert dev.flow_supported() == 2
    dev.close()
    out = []
    for order in (None, 'stages', scrambled, np.arange(n)[::-1]):
        dev = _device(mesh, bath, 0.05)
        m = mesh.boundary_markers
        dev.set_bc(m[0], {'elev': 0.1})
        dev.set_state(uv, eta)
        ends = [n]*9
        if isinstance(order, str):
            _by_stage(dev, ends)
        else:
            if order is not None:
                dev.flow_set_order(order)
            dev.solve_flow(ends)
        out.append(dev.get_state())
        assert dev.flow_timeouts() == 0
        dev.close()
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1])


def test_flow_launch_on_the_ranges_of_a_partition(hip_lib):
    """One rank's cells of a strip partition with a six-layer halo (two time steps between exchanges): owned cells in the
    device's tile order, ghost layers appended, the stage ranges of partition.py."""
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.partition import build_partition, strip_owner
    mesh, bath, uv, eta = channel_case(nx=160, ny=48, seed=9)
    part = build_partition(mesh, strip_owner(mesh, 4), 1, halo_depth=6)
    g = part.local_to_global
    ends = [part.stage_range(s, depth=6) for s in range(6)]
    out = []
    for flow in (False, True):
        dev = Swe2dDevice(part, np.asarray(bath)[part.vertex_global], 0.05, n_owned=part.n_owned, boundary_len=part.boundary_len,
                          ranges=part.reorder_ranges())
        dev.set_state(uv[g], eta[g])
        if flow:
            # ghost cells next to the owned cells they touch (in the device numbering - ghost layers appended layer by layer -
            # a block of ghost cells may have more rim facets than the kernel's staging area holds)
            from thetis_amd import ordering
            dev.flow_set_order(ordering.auto_cell_order(part, 0, part.num_cells))
            assert dev.flow_supported() == 2
            dev.solve_flow(ends)
        else:
            _by_stage(dev, ends)
        out.append(dev.get_state())
        assert dev.flow_timeouts() == 0
        dev.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_advance_takes_the_flow_path_and_matches_the_stage_by_stage_path(hip_lib, monkeypatch):
    mesh, bath, uv, eta = channel_case(nx=41, ny=23, seed=3)
    res = []
    for flag in ('0', '1'):
        monkeypatch.setenv('THETIS_AMD_FLOW', flag)
        monkeypatch.setenv('THETIS_AMD_FUSED_STEP', '0')
        dev = _device(mesh, bath, 0.05)
        dev.set_state(uv, eta)
        dev.advance(37)             # one launch (at most 128 steps per launch)
        dev.advance(263)            # three launches: the stage counters carry over
        res.append(dev.get_state() + (dev.diagnostics(),))
        dev.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_flow_is_refused_where_it_does_not_apply(hip_lib, monkeypatch):
    from thetis_amd._lib import Swe2dError
    mesh, bath, uv, eta = channel_case(nx=9, ny=5, seed=1)
    dev = _device(mesh, bath - 0.6*bath.max(), 0.05)
    dev.set_wetting_and_drying(0.5)                  # wetting-drying is covered since round 5 ...
    assert dev.flow_supported()
    dev.set_viscosity(1.0)                           # ... viscosity is not
    assert not dev.flow_supported()
    with pytest.raises(Swe2dError):
        dev.solve_flow([dev.n_cells]*3)
    dev.close()
    dev = _device(mesh, bath, 0.05)
    for bad in ([dev.n_cells]*4, [10, 20, 20], [dev.n_cells + 1]*3, []):
        with pytest.raises(Swe2dError):
            dev.solve_flow(bad)
    dev.close()
    # more blocks than the device holds resident at once: refused, not deadlocked
    monkeypatch.setenv('THETIS_AMD_FLOW_CAPACITY', '16')
    mesh, bath, uv, eta = channel_case(nx=40, ny=20, seed=1)        # 1600 cells = 25 blocks
    dev = _device(mesh, bath, 0.05)
    assert dev.flow_supported() == 0
    with pytest.raises(Swe2dError):
        dev.solve_flow([dev.n_cells]*3)
    dev.close()


def test_a_block_that_never_arrives_costs_a_bounded_wait_and_is_reported(hip_lib, monkeypatch):
    """Residency cannot be broken on purpose from here, so the flags are: a launch whose stage counters start out of step
    (one block's counter is behind: its neighbours wait for a stage it believes it has already published) ends after the
    bounded wait, and the next synchronisation point reports it."""
    import ctypes
    from thetis_amd._lib import Swe2dError
    monkeypatch.setenv('THETIS_AMD_FLOW_TIMEOUT_S', '0.05')
    mesh, bath, uv, eta = channel_case(nx=40, ny=20, seed=1)
    dev = _device(mesh, bath, 0.05)
    dev.set_state(uv, eta)
    dev.solve_flow([dev.n_cells]*3)
    dev.synchronize()
    assert dev.lib.swe2d_debug_flow_poke(dev.h, 3, 1000) == 0        # block 3 starts the next launch 1000 stages "ahead"
    dev.solve_flow([dev.n_cells]*3)
    assert dev.flow_timeouts() > 0
    with pytest.raises(Swe2dError, match='timed out'):
        dev.synchronize()
    # the handle is usable again
    dev.set_state(uv, eta)
    dev.solve_flow([dev.n_cells]*3)
    dev.synchronize()
    a = dev.get_state()
    dev.set_state(uv, eta)
    _by_stage(dev, [dev.n_cells]*3)
    b = dev.get_state()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    dev.close()


def _delay_build(dev):
    """True when the loaded library is the -DSWE_FLOW_DELAY build (tools/range_check.sh runs these tests against it)"""
    from thetis_amd import _lib
    rc = dev.lib.swe2d_debug_flow_delay(dev.h, -1, 0, 0, 1)
    return rc == _lib.OK


@pytest.mark.parametrize('where,us,every', [(1, 15, 1), (2, 15, 1), (3, 12, 2), (1, 18, 3), (2, 18, 5)])
@pytest.mark.parametrize('case', ['channel', 'shrinking'])
def test_a_block_that_lags_two_to_three_stage_periods_changes_no_bit(hip_lib, case, where, us, every):
    """Adversary for the granule protocol (-DSWE_FLOW_DELAY build only; skipped with the product library): one block sleeps
    12-18 us - two to three stage periods - before its polling pass (1), before it publishes (2) or both (3), in every / every n-th
    stage, while its neighbours run as far ahead as the two slot parities let them.  'shrinking': one rank's cells of a strip
    partition on the twelve shrinking stage ranges of a halo cycle, twice - rim cells drop out of the range while the block that
    faces them lags (their slots must keep the value of their last active stage).  Bit for bit the stage launches."""
    from thetis_amd.device import Swe2dDevice
    if case == 'shrinking':
        from thetis_amd import ordering
        from thetis_amd.partition import build_partition, strip_owner
        mesh, bath, uv, eta = channel_case(nx=160, ny=48, seed=9)
        part = build_partition(mesh, strip_owner(mesh, 4), 1, halo_depth=12)
        g = part.local_to_global
        dev = Swe2dDevice(part, np.asarray(bath)[part.vertex_global], 0.05, n_owned=part.n_owned, boundary_len=part.boundary_len,
                          ranges=part.reorder_ranges())
        dev.flow_set_order(ordering.auto_cell_order(part, 0, part.num_cells))
        uv, eta = uv[g], eta[g]
        ends = [part.stage_range(s_, depth=12) for s_ in range(12)]
        launches = [ends, ends]
    else:
        mesh, bath, uv, eta = channel_case(nx=67, ny=31, seed=11)
        dev = _device(mesh, bath, 0.05)
        launches = [[dev.n_cells]*12, [dev.n_cells]*18]
    if not _delay_build(dev):
        dev.close()
        pytest.skip('needs the -DSWE_FLOW_DELAY build (tools/range_check.sh)')
    n_blocks = (dev.n_cells + 63)//64
    dev.set_state(uv, eta)
    for ends in launches:
        _by_stage(dev, ends)
    ref = dev.get_state()
    for blk in (n_blocks//2, 1, n_blocks - 2, n_blocks//3):
        dev.set_state(uv, eta)
        assert dev.lib.swe2d_debug_flow_delay(dev.h, blk, where, us, every) == 0
        for ends in launches:
            dev.solve_flow(ends)
        assert dev.flow_timeouts() == 0
        got = dev.get_state()
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), 'block {:d}'.format(blk)
    dev.lib.swe2d_debug_flow_delay(dev.h, -1, 0, 0, 1)
    dev.close()


def _tear_build(dev):
    """None with the product library; else whether the -DSWE_FLOW_TEAR build's consumers test the granules' check word (True) or not
    (False: -DSWE_FLOW_NOCHECK, the negative control)"""
    rc = dev.lib.swe2d_debug_flow_tear(dev.h, -3, 0, 1, 0)
    return None if rc < 0 else bool(rc)


@pytest.mark.parametrize('every', [1, 3])
def test_granule_stores_that_land_in_two_halves_change_no_bit(hip_lib, every):
    """Adversary for the 16-byte granules (-DSWE_FLOW_TEAR build only; skipped with the product library): every block stores the
    half of a granule that carries the NEW tag first and the value it belongs to 3 us later - what a store torn on its way to another
    XCD or another device would look like to a consumer polling in between.  The check word (tag ^ lo ^ hi of the value) makes the
    consumer re-poll such a granule: bit for bit the stage launches.  With the negative-control build (-DSWE_FLOW_NOCHECK: the consumer
    looks at the tag only, the protocol of rounds 3-4) the same run MUST give other bits - the test then asserts that it does."""
    mesh, bath, uv, eta = channel_case(nx=67, ny=31, seed=11)
    dev = _device(mesh, bath, 0.05)
    checked = _tear_build(dev)
    if checked is None:
        dev.close()
        pytest.skip('needs the -DSWE_FLOW_TEAR build (tools/range_check.sh)')
    launches = [[dev.n_cells]*12, [dev.n_cells]*18]
    dev.set_state(uv, eta)
    for ends in launches:
        _by_stage(dev, ends)
    ref = dev.get_state()
    dev.set_state(uv, eta)
    assert dev.lib.swe2d_debug_flow_tear(dev.h, -2, 3, every, 0) == 0
    for ends in launches:
        dev.solve_flow(ends)
    assert dev.flow_timeouts() == 0
    got = dev.get_state()
    dev.lib.swe2d_debug_flow_tear(dev.h, -1, 0, 1, 0)
    dev.close()
    same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    if checked:
        assert same
    else:
        assert not same, 'negative control: torn granules taken by their tag alone must show up as wrong bits'


def _beach(nx, ny, seed):
    """A sloping beach with dry ground at rest level (the Balzano geometry in small): bed -h from -1.2 m (sea) to +0.6 m (land)."""
    from thetis_amd.mesh import RectangleMesh
    mesh = RectangleMesh(nx, ny, 13800.0, 7200.0)
    x, y = mesh.vertex_xy.T
    bath = 1.2 - 1.8*x/13800.0 + 0.05*np.sin(y/900.0)
    rng = np.random.default_rng(seed)
    cxy = mesh.cell_xy()
    eta = 0.1*np.cos(cxy[:, :, 0]/4000.0) + 0.02*rng.uniform(-1, 1, size=(mesh.num_cells, 3))
    uv = 0.05*rng.uniform(-1, 1, size=(mesh.num_cells, 3, 2))
    return mesh, bath, uv, eta


@pytest.mark.parametrize('case', ['plain', 'manning_tide', 'no_lf', 'sources_large', 'ranges'])
def test_flow_launch_with_wetting_and_drying_gives_the_bits_of_the_stage_launches(hip_lib, case, monkeypatch):
    """swe_flow_kernel<..., WD> (round 5): the block's values, rim granules and U(0) are the displaced depth D the planes hold, the
    elevation is recovered per stage, the stage ends with the positivity limiter and the dry-ground relaxation - bit for bit
    swe_stage_kernel<true, LF, ., SRC, true> on a beach with dry cells, with Manning friction and a tidal boundary, on shrinking
    ranges, over several launches; swe2d_advance takes the kernel by itself and THETIS_AMD_FLOW_WD=0 keeps it away."""
    from thetis_amd import _lib
    nx, ny = (250, 125) if case == 'sources_large' else (61, 29)
    mesh, bath, uv, eta = _beach(nx, ny, seed=5)
    kw = {'use_lax_friedrichs_velocity': False} if case == 'no_lf' else {}
    n_steps = 6
    out = []
    for flow in (False, True, 'advance'):
        if case == 'ranges':
            kw['reorder'] = None                       # natural numbering: a cell's neighbours lie within one mesh row of 2 nx cells
        dev = _device(mesh, bath, 0.4 if nx < 100 else 0.1, **kw)
        dev.set_wetting_and_drying(0.4)
        if case in ('manning_tide', 'sources_large', 'ranges'):
            dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
            dev.set_bc(mesh.boundary_markers[0], {'elev': -0.3})
        if case == 'sources_large':
            dev.set_field(_lib.FIELD_CORIOLIS, 1e-4*np.ones((mesh.num_cells, 3)))
        assert dev.flow_supported() == (2 if case in ('plain', 'no_lf') else 1)
        dev.set_state(uv, eta)
        n = dev.n_cells
        if case == 'ranges':
            ends = [n - 2*nx*s for s in range(12)]                             # the shrinking ranges of an exchange cycle
        else:
            ends = [n]*(3*n_steps)
        if flow == 'advance':
            if case == 'ranges':
                dev.close()
                continue
            dev.advance(n_steps)                                               # picks the flow kernel by itself
        elif flow:
            if case == 'ranges':
                dev.solve_flow(ends)
            else:
                dev.solve_flow(ends[:6])
                dev.solve_flow(ends[6:])
        else:
            _by_stage(dev, ends)
        out.append(dev.get_state())
        assert dev.flow_timeouts() == 0
        dev.close()
    assert np.isfinite(out[0][0]).all() and np.isfinite(out[0][1]).all()
    depth = out[0][1] + bath[mesh.cells]
    assert (depth < 0.0).any() and (depth > 0.5).any()                          # dry land and open water in the same mesh
    for o in out[1:]:
        assert np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1])
    if case == 'plain':
        monkeypatch.setenv('THETIS_AMD_FLOW_WD', '0')
        dev = _device(mesh, bath, 0.4)
        dev.set_wetting_and_drying(0.4)
        assert dev.flow_supported() == 0
        dev.close()
