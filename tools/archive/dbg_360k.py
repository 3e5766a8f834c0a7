"""debug: is the 360 k-cell channel case of tests/dist_worker.py stable at dt = 2 s at all?  (single device, stage launches / default)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dist_worker
from thetis_amd.device import Swe2dDevice
from thetis_amd import _lib
dist_worker.CASE = 'channel360k'
mesh, bath, uv, eta = dist_worker._case()
for mode in (0, None):
    for dt in (2.0, 0.5):
        dev = Swe2dDevice(mesh, bath, dt)
        dev.set_option(_lib.OPT_FUSED_STAGES, mode)
        dev.set_state(uv, eta)
        try:
            for i in range(16):
                dev.advance(1)
                u, e = dev.get_state()
                if not np.isfinite(e).all():
                    print('mode', mode, 'dt', dt, 'non-finite after step', i + 1); break
            else:
                print('mode', mode, 'dt', dt, 'finite after 16 steps, max |eta|', np.abs(e).max(), 'triple', dev.fused_triple_info()[0])
        except Exception as ex:
            print('mode', mode, 'dt', dt, 'error', ex)
        dev.close()
