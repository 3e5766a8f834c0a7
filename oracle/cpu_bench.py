"""
ORACLE - TEST INFRASTRUCTURE.  bench.py's ``cpu_baseline`` leg: times oracle/swe2d_ref.c (swe2d_ref_advance_blocked) on the
host cores in a process of its own, so that the OpenMP runtime starts with the binding asked for (PyTorch brings its own
OpenMP runtime, already initialised in the bench process).  Prints one JSON line; optionally writes the final state.

    OMP_PROC_BIND=close OMP_PLACES=threads python -m oracle.cpu_bench --nx 1000 --ny 500 --budget 10 [--out state.npz]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nx', type=int, default=1000)
    ap.add_argument('--ny', type=int, default=500)
    ap.add_argument('--budget', type=float, default=10.0, help='seconds of stepping for the all-thread figure')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import bench
    from oracle.ref_lib import RefSWE
    mesh, bath, uv, eta = bench.build_case(args.nx, args.ny)
    ref = RefSWE(mesh.cell_xy(), mesh.cell_nbr, mesh.cell_nbr_facet, bath[mesh.cells], boundary_len=mesh.boundary_len)
    n = mesh.num_cells
    nthreads = ref.num_threads()
    _, _, t2 = ref.advance_blocked(uv, eta, bench.DT, 2)                 # warm-up: thread team, page faults of the library
    _, _, t2 = ref.advance_blocked(uv, eta, bench.DT, 2)
    steps = int(max(3, min(200, args.budget/(t2/2.0)/3.0)))
    runs = []
    for _ in range(3):                                                   # median of three multi-step runs
        u_c, e_c, t = ref.advance_blocked(uv, eta, bench.DT, steps)
        runs.append(t)
    t_med = float(np.median(runs))
    # one thread: median of three two-step runs (first touch and the loop on the same core)
    ref.set_num_threads(1)
    one = []
    for _ in range(3):
        _, _, t1 = ref.advance_blocked(uv, eta, bench.DT, 2)
        one.append(t1/2.0)
    ref.set_num_threads(nthreads)
    t1_med = float(np.median(one))
    out = {'value': n*3.0*steps/t_med, 'unit': 'element-updates/s', 'cores': nthreads, 'kind': 'port',
           'value_1core': n*3.0/t1_med, 'speedup_over_1core': t1_med*steps/t_med,
           'steps': steps, 'seconds': runs, 'seconds_per_step_1core': one,
           'omp': {k: os.environ.get(k) for k in ('OMP_PROC_BIND', 'OMP_PLACES', 'OMP_NUM_THREADS')},
           'host_cpus': os.cpu_count()}
    if args.out:
        np.savez(args.out, uv=u_c, eta=e_c, steps=steps)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
