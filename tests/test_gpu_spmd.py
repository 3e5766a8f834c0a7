"""
GPU: an UNCHANGED FlowSolver2d user script under several ranks (one process per rank, here sharing the one GPU of the test box:
gloo control plane, IPC peer-to-peer halos exactly as between GPUs) is domain-decomposed and gives the single-device run bit for
bit - state, iteration / time / export counters, callback histories, exported files.  The reference's counterpart:
``mpiexec -n N python script.py`` (examples/README.md:51-56).  CPU twin: tests/test_spmd.py.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from dist_worker import run_spmd
from test_spmd import _check, single_rank

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name,world', [('channel', 2), ('channel', 4), ('forced', 2), ('forced', 3), ('tracer', 2), ('tracer_forced', 3),
                                        ('tracer_only', 2), ('balzano', 2), ('balzano', 4), ('forced_fe', 2), ('tracer_fe', 2),
                                        ('tracer_nolim', 2), ('fields', 2), ('fields', 3), ('periodic', 2), ('periodic', 4), ('coast', 2), ('coast', 4)])
def test_user_script_under_n_ranks_on_one_gpu(tmp_path, tmp_path_factory, hip_lib, name, world):
    single = single_rank(name, tmp_path_factory, cpu=False)
    ranks = run_spmd(world, str(tmp_path), name, cpu=False)
    assert ranks[0]['exchange'] == 'p2p'            # the default transport set up (hipIpc* between the rank processes)
    _check(single, ranks, exact_callbacks=True)


@pytest.mark.parametrize('name,world,env', [
    ('channel', 3, {'THETIS_AMD_SPMD_FLOW': '1'}),                                    # dataflow launches, exchange inside (FX)
    ('channel', 2, {'THETIS_AMD_SPMD_FLOW': '1', 'THETIS_AMD_EXCHANGE_EVERY': '1'}),
    ('forced', 2, {'THETIS_AMD_SPMD_FLOW': '1'}),                                     # stage by stage between flow-capable batches
    ('channel', 2, {'THETIS_AMD_EXCHANGE': 'host'}), ('tracer', 2, {'THETIS_AMD_EXCHANGE': 'host', 'THETIS_AMD_EXCHANGE_EVERY': '1'}),
    ('channel', 4, {'THETIS_AMD_PARTITION': 'rcb', 'THETIS_AMD_EXCHANGE_EVERY': '1'}),
    ('tracer', 2, {'THETIS_AMD_OVERLAP_STAGES': '2'}),
    ('channel', 8, {}),
    ('restart', 3, {}),
    # the ranks DISAGREE on whether the dataflow kernel covers their partition (capacity forced to 16 blocks: the end ranks of three,
    # with one ghost side, fit; the middle rank does not): the common decision must be reached without leaving anyone in a collective
    ('channel_wide', 3, {'THETIS_AMD_FLOW_CAPACITY': '16'}),
])
def test_user_script_variants_on_one_gpu(tmp_path, tmp_path_factory, hip_lib, name, world, env):
    single = single_rank(name, tmp_path_factory, cpu=False)
    ranks = run_spmd(world, str(tmp_path), name, cpu=False, env=env)
    _check(single, ranks, exact_callbacks=True)


@pytest.mark.parametrize('world', [2, 8])
def test_full_size_channel_under_ranks_matches_single_device(tmp_path, tmp_path_factory, hip_lib, world):
    """BASELINE cfg 3's mesh size with the bits checked: the channel2d script on 1 M triangles (8 000 automatic time steps, five
    print_state / volume-check points), one device against 2 and 8 ranks sharing it - every cell of the final state, the step and
    export counters and the callback histories identical.  (Eight strips of 125 000 cells: the end ranks would fit the dataflow
    kernel, the middle ranks with their two ghost sides would not - the ranks' common decision, and they share a GPU: stage launches.)"""
    single = single_rank('channel_1m', tmp_path_factory, cpu=False)
    ranks = run_spmd(world, str(tmp_path), 'channel_1m', cpu=False, timeout=900)
    _check(single, ranks, exact_callbacks=True)
    assert single[0]['elev'].shape == (3000000,) and single[0]['iteration'] > 5000


def test_a_transport_that_cannot_be_set_up_is_left_by_all_ranks_together(tmp_path, tmp_path_factory, hip_lib):
    """the peer-to-peer zones cannot be mapped (THETIS_AMD_TEST_BREAK_P2P makes swe2d_p2p_open fail, as on a node without IPC peer
    access): every rank gives the transport up in the same all-reduce and the run goes through host memory - same bits"""
    single = single_rank('forced', tmp_path_factory, cpu=False)
    ranks = run_spmd(3, str(tmp_path), 'forced', cpu=False, env={'THETIS_AMD_TEST_BREAK_P2P': '1'})
    assert [r['exchange'] for r in ranks] == ['host']*3
    _check(single, ranks, exact_callbacks=True)


def test_a_rank_whose_handle_cannot_be_built_takes_its_peers_out_of_the_handshake(tmp_path, tmp_path_factory, hip_lib):
    """rank 1 fails BEFORE the peer-to-peer handshake (its first handle cannot be built); it still enters the
    handshake's all-gather with its error, every rank leaves it with the same exception, agrees on the failure and the run goes
    through host memory - nobody is left waiting in a mismatched collective until the gloo timeout.  Same bits."""
    single = single_rank('forced', tmp_path_factory, cpu=False)
    ranks = run_spmd(3, str(tmp_path), 'forced', cpu=False, env={'THETIS_AMD_TEST_FAIL_HANDLE_RANK': '1'}, timeout=240)
    assert [r['exchange'] for r in ranks] == ['host']*3
    _check(single, ranks, exact_callbacks=True)


def test_periodic_verification_of_the_in_launch_exchange(tmp_path, tmp_path_factory, hip_lib):
    """THETIS_AMD_VERIFY_EVERY on the GPU: the channel script under three ranks with the dataflow launches and the exchange inside
    them, every window of eight steps replayed with stage launches through host memory.  Product library: every window agrees.
    -DSWE_FLOW_TEAR build (tools/range_check.sh; granule stores in two halves): the same with the check word; with its negative
    control (-DSWE_FLOW_NOCHECK: torn granules are taken by their tag) the verification MUST report mismatches, keep the replayed
    state, drop the dataflow launches - and the run still ends on the single-device bits."""
    from thetis_amd.device import Swe2dDevice
    from thetis_amd.mesh import RectangleMesh
    m = RectangleMesh(4, 2, 1.0, 1.0)
    dev = Swe2dDevice(m, np.ones(m.num_vertices), 1e-3)
    kind = dev.lib.swe2d_debug_flow_tear(dev.h, -3, 0, 1, 0)          # < 0 product build, 1 tear + check word, 0 tear without
    dev.close()
    env = {'THETIS_AMD_SPMD_FLOW': '1', 'THETIS_AMD_VERIFY_EVERY': '8'}
    if kind >= 0:
        env['THETIS_AMD_TEST_TEAR'] = '3:1'
    single = single_rank('channel', tmp_path_factory, cpu=False)
    ranks = run_spmd(3, str(tmp_path), 'channel', cpu=False, env=env)
    _check(single, ranks, exact_callbacks=True)
    for r in ranks:
        rep = r['verify_report']
        assert rep['windows'] >= 3
        assert (rep['mismatches'] >= 1) if kind == 0 else (rep['mismatches'] == 0), rep


def _script(args, world, port):
    e = dict(os.environ)
    e['THETIS_AMD_DIST_BACKEND'] = 'gloo'            # the ranks share the one GPU of the test box: RCCL would refuse them
    cmd = [sys.executable] + args if world == 1 else \
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
         '--master-port', str(port)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize('script,args,key,world', [
    ('channel2d.py', ['--t-end', '200'], 'volume ', 2), ('channel2d.py', ['--t-end', '200'], 'volume ', 3),
    ('balzano.py', ['--hours', '1'], 'finite ', 2), ('tracer2d.py', ['--revolutions', '0.25', '--limiter'], 'relative L2', 2)])
def test_example_scripts_under_torch_distributed_run(hip_lib, script, args, key, world):
    """the files under examples/ as they are: ``python -m torch.distributed.run --nproc-per-node N examples/x.py`` prints, on every
    rank, the result line of ``python examples/x.py``; the solver's own output (print_state, callbacks) appears once (rank 0)"""
    path = os.path.join('examples', script)
    one = _script([path] + args, 1, 0)
    many = _script([path] + args, world, 29640 + world)
    ref = [l for l in one.splitlines() if l.startswith(key)]
    # (the ranks print at the same moment: their lines may share a line of the captured output)
    assert len(ref) == 1 and many.count(ref[0]) == world and many.count(key) == world, (ref, [l for l in many.splitlines() if key in l])
    # print_state lines (exp iter time norms... Tcpu): once, and the same numbers up to the wall-clock column
    state = lambda out: [l.split()[:-1] for l in out.splitlines() if len(l.split()) >= 5 and l.split()[0].isdigit() and l.split()[1].isdigit()]
    assert state(many) == state(one) and len(state(one)) >= 2
