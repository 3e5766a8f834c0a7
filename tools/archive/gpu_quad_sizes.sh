#!/bin/bash
# size sweep of the quadrilateral SWE step: one lane per cell / two lanes per cell
O=gpurun_out/quad2; mkdir -p $O
run() { python - "$@" <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import cfgbench
from thetis_amd.device import Swe2dDevice
from thetis_amd.mesh import RectangleMesh
rng = np.random.default_rng(1234)
for nx in [int(a) for a in sys.argv[1:]]:
    m = RectangleMesh(nx, nx, 100e3, 100e3, quadrilateral=True)
    c = m.cell_xy()
    eta = 0.5*np.exp(-((c[:, :, 0] - 50e3)**2 + (c[:, :, 1] - 50e3)**2)/(5e3)**2)
    dev = Swe2dDevice(m, np.full(m.num_vertices, 20.0), 0.25)
    dev.set_state(1e-3*rng.uniform(-1, 1, size=(m.num_cells, 4, 2)), eta)
    t = cfgbench.timed(dev, dev.advance, 50)
    print('%-20s lanes %-2s %8d cells %8.1f us/step  frac %.3f' % (os.environ.get('THETIS_AMD_LIB', 'default')[-20:], os.environ.get('THETIS_AMD_QUAD_LANES', '2'), m.num_cells, 1e6*t, 936.0*m.num_cells/t/8e12))
    dev.close()
PY
}
( THETIS_AMD_QUAD_LANES=1 run 350 500 700 850 1000 1400
  THETIS_AMD_LIB=$PWD/variants/q2_int.so run 350 500 700 850 1000 1400 ) 2>&1 | grep -v amdgpu.ids | tee $O/sizes.txt
