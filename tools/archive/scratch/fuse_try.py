import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
from helpers import channel_case
from thetis_amd.device import Swe2dDevice
os.environ['THETIS_AMD_FLOW'] = '0'
def run(mesh, bath, uv, eta, fuse, steps, reorder='auto', bc=True):
    os.environ['THETIS_AMD_FUSE12'] = fuse
    dev = Swe2dDevice(mesh, bath, 0.5, reorder=reorder)
    if bc: dev.set_bc(2, {'elev': 0.1})
    dev.set_state(uv, eta)
    dev.advance(steps)
    out = dev.get_state()
    dev.close()
    return out
for (nx, ny) in ((200, 120), (53, 31)):
    mesh, bath, uv, eta = channel_case(nx=nx, ny=ny, lx=100e3, ly=50e3, seed=5, amp_eta=0.3, amp_u=0.2)
    a = run(mesh, bath, uv, eta, '0', 5); b = run(mesh, bath, uv, eta, '1', 5)
    print(nx, ny, 'bitwise', np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), 'finite', np.isfinite(a[1]).all(), 'maxdiff', np.abs(a[1]-b[1]).max())
# timing at 1 M cells
import bench
mesh, bath, uv, eta = bench.build_case()
for fuse in ('0', '1', '0', '1'):
    os.environ['THETIS_AMD_FUSE12'] = fuse
    dev = Swe2dDevice(mesh, bath, 0.25)
    dev.set_state(uv, eta)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        dev.advance(20); dev.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); dev.advance(100); dev.synchronize(); best = min(best, (time.perf_counter() - t0)/100)
    st = dev.get_state()
    print('fuse', fuse, 'us/step', 1e6*best, 'checksum', float(np.abs(st[1]).sum()))
    dev.close()
