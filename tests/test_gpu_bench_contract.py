"""GPU: bench.py keeps its contract - one JSON line (the LAST stdout line) with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('env', [{}, {'THETIS_AMD_FORCE_DIST': '1'},
                                 {'THETIS_AMD_FORCE_DIST': '1', 'THETIS_AMD_TUNE_SCHEDULE': '1'}],
                         ids=['single', 'distributed_path_world1', 'distributed_path_world1_schedule_tuning'])
def test_bench_prints_one_json_line_with_the_contract_fields(hip_lib, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-cpu', '--prewarm', '0.05']
                       + (['--no-beyond-cache'] if env else []),
                       capture_output=True, text=True, env=e, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                      # the JSON is the last stdout line (RCCL banner flushed before it)
    assert sum(1 for l in lines if l.lstrip().startswith('{"metric"')) == 1
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in d, key
    assert d['steps'] == 6 and d['warmup'] == 2 and d['n_gpus'] == 1 and d['dtype'] == 'f64' and d['vs_baseline'] is None
    assert d['unit'] == 'element-updates/s' and d['higher_is_better'] is True and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['peak'] == 8000.0 and rf['unit'] == 'GB/s'
    assert abs(rf['frac'] - rf['achieved']/rf['peak']) < 1e-12
    # whole-job throughput = cells * 3 stages * steps / time
    assert abs(d['value'] - 1e6*3*6/(d['ms_per_step']*6e-3))/d['value'] < 1e-9
    assert 1e9 < d['value'] < 1e11
    if not env:
        # the line carries its own spread: five K-step regions back to back, the headline figures are those of the first
        assert len(rf['frac_samples']) == 5 and rf['frac_samples'][0] == rf['frac']
        assert rf['frac_min'] <= rf['frac_median'] <= rf['frac_max'] and rf['frac_min'] == min(rf['frac_samples'])
        # the same kernel beyond the Infinity Cache (4M triangles), reported next to the headline fraction
        assert 0.05 < rf['frac_beyond_cache'] < 1.0 and rf['beyond_cache']['algorithmic_bytes_per_launch'] == 228.0*4e6
    if env.get('THETIS_AMD_TUNE_SCHEDULE'):
        tuned = d['config']['schedule_tuning']
        assert len(tuned) >= 5 and all(t['us_per_step'] > 0 for t in tuned)
        best = min(tuned, key=lambda t: t['us_per_step'])
        assert (d['config']['exchange'], d['config']['exchange_every'], d['config']['overlap_stages']) == (
            best['exchange'], best['exchange_every'], best['overlap_stages'])
        assert d['config']['volume_conserved'] is True
    if env:
        assert d['config']['failures'] == [] and 'p2p' in d['config']['transports_verified']


def test_bench_two_ranks_through_torch_distributed_run(hip_lib):
    """The driver's N > 1 invocation (python -m torch.distributed.run ... bench.py --gpus N) with two ranks sharing the GPU
    of the test box: RCCL refuses two ranks on one device, so THETIS_AMD_DIST_BACKEND=gloo drops the 'rccl' transport; the
    peer-to-peer transport (IPC-mapped landing zones, the default choice on a multi-GPU node) is verified bit for bit
    against the host-staged exchange and timed, and everything else - partitioning, schedule tuning with the max over
    ranks, graph capture, the timed region, rank 0 printing one JSON line - is the code a multi-GPU node runs."""
    e = dict(os.environ)
    e.update({'THETIS_AMD_DIST_BACKEND': 'gloo', 'THETIS_AMD_LARGE_MESH': '1600,400', 'THETIS_AMD_SOAK_S': '0.5'})
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29577', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '16',
                        '--warmup', '2', '--prewarm', '0.05'],
                       capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 16 and d['scaling'] == 'strong'
    cfg = d['config']
    assert len(cfg['schedule_tuning']) >= 5 and cfg['volume_conserved'] is True
    assert cfg['transports_verified'] == ['p2p', 'host'] and cfg['exchange'] == 'p2p' and cfg['p2p_timeouts'] == 0
    assert cfg['failures'] == [] and any(t['graph_mode'] != 'none' for t in cfg['schedule_tuning'])
    assert 1e8 < d['value'] < 1e11
    # SURVEY 8(d) cfg 3: the exchange's share of the step, measured after the timed region (the same steps without sends / receives)
    assert cfg['exchange_time_fraction'] is not None and 0.0 <= cfg['exchange_time_fraction'] < 0.9, cfg['exchange_time_fraction_note']
    # set-up (transport checks + schedule tuning) is bounded: candidates beyond the budget are skipped and listed
    assert cfg['setup_s'] <= cfg['setup_budget_s'] + 15.0 and cfg['setup_budget_s'] == 60.0 and isinstance(cfg['setup_skipped'], list)
    # the second, untuned timed region on the larger mesh of the same channel (here shrunk: 2 ranks share one GPU)
    lm = cfg['large_mesh']
    assert lm['n_cells'] == 2*1600*400 and lm['volume_conserved'] is True and lm['value'] > 1e8 and lm['speedup_model'] > 0
    # the soak (round 5): the peer-to-peer transport stepped >= 0.5 s from the initial state and ended on the bits of the whole mesh
    # stepped by one GPU alone
    sk = cfg['soak']
    assert sk['steps'] >= 240 and any(v['what'] == "transport 'p2p'" and v['seconds'] >= 0.4 for v in sk['verified']), sk


def test_bench_eight_ranks_through_torch_distributed_run(hip_lib):
    """BASELINE cfg 3's rank count, first contact rehearsed: ``--gpus 8`` with eight ranks sharing the test GPU (gloo control
    plane, IPC peer-to-peer halos among eight processes, middle ranks with two peers) on a shrunk mesh of the same channel
    (THETIS_AMD_BENCH_MESH; eight processes on the full mesh would only test the box's patience) and a shrunk large-mesh region.
    The run must end in ONE complete JSON line with ``config.large_mesh``."""
    e = dict(os.environ)
    e.update({'THETIS_AMD_DIST_BACKEND': 'gloo', 'THETIS_AMD_BENCH_MESH': '256,64', 'THETIS_AMD_LARGE_MESH': '512,64',
              'THETIS_AMD_SETUP_BUDGET_S': '40', 'THETIS_AMD_SOAK_S': '0.3'})
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr',
                        '127.0.0.1', '--master-port', '29581', os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '16',
                        '--warmup', '2', '--prewarm', '0.05'],
                       capture_output=True, text=True, env=e, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cfg = d['config']
    assert d['n_gpus'] == 8 and d['steps'] == 16 and d['scaling'] == 'strong' and d['value'] > 1e6
    assert cfg['n_cells'] == 2*256*64 and 'NOT the BASELINE workload' in cfg['workload']
    assert cfg['transports_verified'][0] == 'p2p' and cfg['exchange'] == 'p2p' and cfg['p2p_timeouts'] == 0 and cfg['flow_timeouts'] == 0
    assert cfg['volume_conserved'] is True and len(cfg['schedule_tuning']) >= 1
    lm = cfg['large_mesh']
    assert lm is not None and lm['n_cells'] == 2*512*64 and lm['volume_conserved'] is True


def test_first_contact_tells_the_story_of_a_multi_rank_run(hip_lib, tmp_path):
    """``python -m tools.first_contact --gpus N``: the bench under torch.distributed.run with its running log passed through and a
    readable summary - transports, soak, every schedule candidate, what was chosen and why, failures - so that a scaling run that
    goes wrong on a node nobody has seen leaves a usable log (VERDICT r05 "next 4").  Here: four ranks sharing the test GPU on a
    shrunk mesh, once as it is and once with the peer-to-peer mapping broken (the failure must be in the story)."""
    for broken in (False, True):
        e = dict(os.environ)
        e.update({'THETIS_AMD_LARGE_MESH': '512,64', 'THETIS_AMD_SETUP_BUDGET_S': '25', 'THETIS_AMD_SOAK_S': '0.3'})
        if broken:
            e['THETIS_AMD_TEST_BREAK_P2P'] = '1'
        log = str(tmp_path/('fc{:d}.log'.format(int(broken))))
        r = subprocess.run([sys.executable, '-m', 'tools.first_contact', '--gpus', '4', '--same-gpu', '--mesh', '256,64', '--steps', '16',
                            '--warmup', '2', '--port', str(29590 + int(broken)), '--log', log],
                           capture_output=True, text=True, env=e, timeout=1200, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
        text = open(log).read()
        assert text == r.stdout
        for must in ('4 ranks, control plane gloo', 'result: ', 'transports verified', 'chosen: exchange', 'failures:', 'timed region:'):
            assert must in text, must
        d = json.loads(text.splitlines()[-1])
        assert d['n_gpus'] == 4 and d['config']['volume_conserved'] is True
        if broken:
            assert d['config']['exchange'] == 'host' and "FAILED / dropped: transport 'p2p'" in text and "  - transport 'p2p'" in text
        else:
            assert d['config']['exchange'] == 'p2p' and 'failures: none' in text and 'schedule candidates' in text, \
                [l for l in text.splitlines() if 'FAILED' in l or l.startswith('  - ')]
            assert "transport 'p2p': set up on every rank" in text and 'soak: ' in text


def test_bench_set_up_budget_cuts_the_candidate_list_short(hip_lib):
    """THETIS_AMD_SETUP_BUDGET_S = 0: after the first transport and the first candidate nothing more is tried; the line is
    still complete and says what was skipped."""
    e = dict(os.environ)
    e.update({'THETIS_AMD_DIST_BACKEND': 'gloo', 'THETIS_AMD_SETUP_BUDGET_S': '0', 'THETIS_AMD_NO_LARGE_MESH': '1', 'THETIS_AMD_SOAK_S': '0'})
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29579', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '8',
                        '--warmup', '2', '--prewarm', '0.05'],
                       capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.lstrip().startswith('{"metric"')][0])
    cfg = d['config']
    assert len(cfg['schedule_tuning']) == 1 and len(cfg['setup_skipped']) >= 5 and cfg['volume_conserved'] is True
    assert 'large_mesh' not in cfg and d['value'] > 1e8


def test_bench_survives_a_transport_that_fails(hip_lib):
    """First contact with a node where a transport is broken must still end with the JSON line: the peer-to-peer zone
    cannot be opened (THETIS_AMD_TEST_BREAK_P2P makes swe2d_p2p_open fail), the bench records the failure and falls back."""
    e = dict(os.environ)
    e.update({'THETIS_AMD_DIST_BACKEND': 'gloo', 'THETIS_AMD_TEST_BREAK_P2P': '1'})
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29578', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '8',
                        '--warmup', '2', '--prewarm', '0.05'],
                       capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['config']['exchange'] == 'host' and d['config']['transports_verified'] == ['host']
    assert any('p2p' in f for f in d['config']['failures']) and d['value'] > 1e7


def test_bench_survives_rccl_refusing_its_ranks(hip_lib):
    """First-contact rehearsal with the REAL nccl backend: two ranks on the one GPU of the test box - RCCL refuses the duplicate
    device ("invalid usage") at its first collective.  The bench must record that, keep the transports that verified (peer-to-peer
    through IPC + the host-staged yardstick) and print its one JSON line from rank 0."""
    base = dict(os.environ)
    base.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29612', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0',
                 'THETIS_AMD_DIST_TIMEOUT_S': '120'})
    base.pop('THETIS_AMD_DIST_BACKEND', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '8', '--warmup', '2', '--prewarm', '0.05']
    procs = []
    for rank in (1, 0):
        e = dict(base)
        e['RANK'] = str(rank)
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[1][1][-2000:]
    lines = [l for l in outs[1][0].splitlines() if l.lstrip().startswith('{"metric"')]           # rank 0
    assert len(lines) == 1 and not any(l.lstrip().startswith('{"metric"') for l in outs[0][0].splitlines())
    d = json.loads(lines[0])
    cfg = d['config']
    assert cfg['transports_verified'] == ['p2p', 'host'] and cfg['exchange'] == 'p2p' and cfg['p2p_timeouts'] == 0
    assert any("transport 'rccl'" in f for f in cfg['failures']) and cfg['volume_conserved'] is True
    assert d['n_gpus'] == 2 and d['value'] > 1e8
