"""
CPU (gloo): an unchanged FlowSolver2d user script under several ranks is domain-decomposed and gives the single-rank result -
the reference's ``mpiexec -n N python script.py`` (examples/README.md:51-56) with ``python -m torch.distributed.run``.  The host
logic under test is the product's (thetis_amd/comm.py, spmd.py, distributed.py, solver2d.py, exporter.py, callback.py); the
arithmetic is the oracle's C restatement behind the device interface (tests/cpu_device.py).  The GPU twin of this file is
tests/test_gpu_spmd.py.
"""
import numpy as np
import pytest

from dist_worker import run_spmd

STATE_KEYS = ('uv', 'elev', 'tracer_2d')
_SINGLE = {}


def single_rank(name, tmp_path_factory, cpu=True):
    """the one-rank run of a scenario (once per test session: several partitioned variants are compared with it)"""
    key = (name, cpu)
    if key not in _SINGLE:
        _SINGLE[key] = run_spmd(1, str(tmp_path_factory.mktemp('single_' + name)), name, cpu=cpu)
    return _SINGLE[key]


def _check(single, ranks, exact_callbacks=True):
    s = single[0]
    for r in ranks:
        for key in ('iteration', 'i_export', 'simulation_time', 'dt'):
            assert r[key] == s[key], key
        for key in STATE_KEYS:
            if key in s:
                assert np.array_equal(r[key], s[key]), '{:} differs from the single-rank run'.format(key)
        assert np.array_equal(r['forcing_times'], s['forcing_times'])
        assert set(r['callbacks']) == set(s['callbacks'])
        for name, h in s['callbacks'].items():
            assert r['callbacks'][name].shape == h.shape and h.shape[0] > 0, name
            if exact_callbacks:
                # the integrals are order-independent limb sums (include/swe2d.h: swe2d_diagnostics_limbs): however the mesh is cut, the
                # printed norms and the conservation checks are the same doubles
                assert np.array_equal(r['callbacks'][name], h), name
            else:
                assert np.allclose(r['callbacks'][name], h, rtol=1e-12, atol=1e-13), name
        assert r['files'] == s['files']             # the same files with the same bytes (rank 0 writes the gathered fields)


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('name', ['channel', 'forced', 'tracer', 'tracer_forced', 'tracer_only'])
def test_user_script_under_n_ranks_equals_single_rank(tmp_path, tmp_path_factory, ref_so, name, world):
    single = single_rank(name, tmp_path_factory)
    ranks = run_spmd(world, str(tmp_path), name)
    _check(single, ranks)
    if name == 'channel':
        assert len(single[0]['files']) == 2*(5 + 1) + 2*5 and single[0]['i_export'] == 4      # vtu + pvd, npz; 5 exports each


@pytest.mark.parametrize('name,world,env', [
    ('forced_fe', 2, {}), ('tracer_fe', 2, {}), ('tracer_nolim', 3, {}),
    ('channel', 2, {'THETIS_AMD_EXCHANGE_EVERY': '1'}), ('tracer', 2, {'THETIS_AMD_EXCHANGE_EVERY': '1'}),
    ('tracer_forced', 3, {'THETIS_AMD_EXCHANGE_EVERY': '1'}), ('channel', 3, {'THETIS_AMD_PARTITION': 'rcb'}),
    ('forced', 2, {'THETIS_AMD_PARTITION': 'strip_y', 'THETIS_AMD_EXCHANGE_EVERY': '3'}),
    ('restart', 2, {}),
    ('periodic', 2, {}), ('periodic', 3, {'THETIS_AMD_EXCHANGE_EVERY': '1'}),
    ('coast', 3, {}),
])
def test_user_script_variants(tmp_path, tmp_path_factory, ref_so, name, world, env):
    single = single_rank(name, tmp_path_factory)
    ranks = run_spmd(world, str(tmp_path), name, env=env)
    _check(single, ranks)


@pytest.mark.parametrize('name,world,fault', [('channel', 2, None), ('channel', 3, '1:1'), ('tracer', 2, '0:0')])
def test_periodic_verification_keeps_the_replayed_state(tmp_path, tmp_path_factory, ref_so, name, world, fault):
    """THETIS_AMD_VERIFY_EVERY = 3: every window of three steps is replayed from its start with stage launches and the exchange
    through host memory, the ranks' blake2b verdicts are gathered.  Without a fault: same bits, every window agrees.  With ONE wrong
    bit planted in one rank's fast result of one window (THETIS_AMD_TEST_VERIFY_FAULT = rank:window): the mismatch is reported
    with that rank, the replayed state is kept, and the run ends on the single-rank bits all the same."""
    single = single_rank(name, tmp_path_factory)
    env = {'THETIS_AMD_VERIFY_EVERY': '3'}
    if fault:
        env['THETIS_AMD_TEST_VERIFY_FAULT'] = fault
    ranks = run_spmd(world, str(tmp_path), name, env=env)
    _check(single, ranks)
    for r in ranks:
        rep = r['verify_report']
        assert rep['windows'] > 2
        if fault:
            assert rep['mismatches'] == 1 and rep['bad_ranks'] == [[int(fault.split(':')[0])]]
        else:
            assert rep['mismatches'] == 0


def test_world_8(tmp_path, tmp_path_factory, ref_so):
    """eight ranks on a channel whose strips (3 columns of cells) are narrower than the six-layer halo: a rank's ghost layers
    reach into its second neighbours"""
    single = single_rank('channel', tmp_path_factory)
    ranks = run_spmd(8, str(tmp_path), 'channel')
    _check(single, ranks)
