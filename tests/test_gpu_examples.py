"""GPU: the runnable scripts under examples/ keep working (each is the thetis_amd version of a reference example)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=ROOT, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_channel2d_example(hip_lib):
    out = _run([os.path.join('examples', 'channel2d.py'), '--t-end', '200'])
    rel = float(out.strip().splitlines()[-1].split()[-1])
    assert abs(rel) < 1e-10                                   # closed channel: volume conserved


def test_tracer2d_example(hip_lib):
    out = _run([os.path.join('examples', 'tracer2d.py'), '--revolutions', '1'])
    err, qmin, qmax = [float(out.strip().splitlines()[-1].split()[i]) for i in (3, 5, 7)]
    assert err < 0.2 and qmin > 0.7 and qmax < 2.3            # unlimited scheme: small over/undershoots at the cylinder
    out = _run([os.path.join('examples', 'tracer2d.py'), '--revolutions', '1', '--limiter'])
    err, qmin, qmax = [float(out.strip().splitlines()[-1].split()[i]) for i in (3, 5, 7)]
    # limited once per step (coupled_timeintegrator_2d.py:102-105): bounds hold up to the within-step over/undershoot of the means
    assert err < 0.2 and qmin > 0.99 and qmax < 2.01


def test_multi_gpu_example_single_rank(hip_lib):
    out = _run([os.path.join('examples', 'multi_gpu.py'), '--nx', '200', '--ny', '100', '--steps', '40'],
               env={'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29533'})
    line = [l for l in out.splitlines() if 'element-updates/s' in l][-1]      # RCCL prints its banner after it at exit
    assert float(line.split()[-1]) < 1e-10


def test_balzano_example(hip_lib):
    out = _run([os.path.join('examples', 'balzano.py'), '--hours', '4'])
    last = [l for l in out.splitlines() if l.startswith('finite')][-1].split()
    assert last[1] == 'True' and float(last[5]) < 0.0 and float(last[9]) < 3.0        # the upper beach fell dry; sane speeds
