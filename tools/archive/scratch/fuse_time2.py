import os, sys, time, numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
from thetis_amd.device import Swe2dDevice
from thetis_amd import ordering
import bench
os.environ['THETIS_AMD_FLOW'] = '0'; os.environ['THETIS_AMD_FUSE_STATS'] = '1'
for (nx, ny) in ((1000, 500),):
    mesh, bath, uv, eta = bench.build_case(nx, ny)
    for (bx, by) in ((16, 8), (16, 6), (12, 8), (24, 4), (8, 12)):
        perm = ordering.structured_tile_order(nx, ny, bx=bx, by=by)
        for fuse in ('0', '1', '0', '1'):
            os.environ['THETIS_AMD_FUSE12'] = fuse
            dev = Swe2dDevice(mesh, bath, 0.25, reorder=perm)
            dev.set_state(uv, eta)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.4:
                dev.advance(20); dev.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); dev.advance(60); dev.synchronize(); best = min(best, (time.perf_counter() - t0)/60)
            print('tiles', bx, by, 'fuse', fuse, 'us/step %.2f' % (1e6*best), flush=True)
            dev.close()
