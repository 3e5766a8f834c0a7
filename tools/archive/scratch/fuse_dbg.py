import os, sys, numpy as np
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from thetis_amd.device import Swe2dDevice
import bench
os.environ['THETIS_AMD_FLOW'] = '0'
for (nx, ny) in ((354, 177), (1000, 500)):
    mesh, bath, uv, eta = bench.build_case(nx, ny)
    for nsteps in (2, 3, 10, 60):
        outs = []
        for fuse in ('0', '1'):
            os.environ['THETIS_AMD_FUSE12'] = fuse
            dev = Swe2dDevice(mesh, bath, 0.25)
            dev.set_state(uv, eta)
            dev.advance(nsteps)
            if nsteps == 60: dev.advance(20); dev.advance(20)
            outs.append(dev.get_state())
            perm = dev.perm
            dev.close()
        d = np.abs(outs[0][1] - outs[1][1]).max(axis=1)
        bad = np.nonzero(d > 0)[0]
        inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
        print(nx, ny, 'steps', nsteps, 'bad cells', len(bad), 'max', d.max() if len(bad) else 0.0, 'device ids', np.sort(inv[bad])[:8], 'sum', float(np.abs(outs[0][1]).sum()), float(np.abs(outs[1][1]).sum()))
