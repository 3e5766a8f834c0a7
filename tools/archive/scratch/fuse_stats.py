import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
from thetis_amd.device import Swe2dDevice
import bench
os.environ['THETIS_AMD_FLOW'] = '0'; os.environ['THETIS_AMD_FUSE12'] = '1'; os.environ['THETIS_AMD_FUSE_STATS'] = '1'
mesh, bath, uv, eta = bench.build_case(1000, 500)
dev = Swe2dDevice(mesh, bath, 0.25); dev.set_state(uv, eta); dev.advance(1); dev.synchronize(); dev.close()
