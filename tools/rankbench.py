#!/usr/bin/env python
"""Per-rank cost of the strip-partitioned bench on ONE GPU: builds rank r's partition of a `world`-rank run and times its
step loop (stage kernels on the shrinking ranges, pack, unpack) with the exchange itself stubbed out (recv buffer =
stale data; the numbers are timing only).  python tools/rankbench.py --world 8 --rank 3 --every 4 --steps 240"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=3)
    ap.add_argument('--every', type=int, default=1)
    ap.add_argument('--steps', type=int, default=240)
    ap.add_argument('--overlap', type=int, default=0)
    ap.add_argument('--prewarm', type=float, default=0.5)
    ap.add_argument('--exchange', default='stub', help="'stub': pack/unpack kernels, no transfer; 'p2p': the peer-to-peer "
                    "kernels pushing into this rank's OWN landing zone (same-device loopback: timing only)")
    ap.add_argument('--nosplit', action='store_true')
    ap.add_argument('--graph-mode', default=None)
    ap.add_argument('--flow', type=int, default=-1, help='1 / 0: the stages of a cycle as one dataflow launch (csrc/swe2d_flow.h) / stage launches (default: the solver decides)')
    ap.add_argument('--flowx', type=int, default=-1, help='1 / 0: the exchange inside the flow launches / separate push and unpack kernels')
    ap.add_argument('--nx', type=int, default=0, help='mesh RectangleMesh(nx, nx/2) instead of the bench mesh')
    ap.add_argument('--case', default='cfg2', help="cfg2: the bench mesh, shallow water only | cfg4: + one tracer with the limiter (coupled cycles, "
                    "combined exchange) | cfg4_tracer_only: the tracer alone (demo_2d_tracer mode) | cfg5: the Balzano geometry at 500 k "
                    "triangles with wetting-drying, Manning friction and the tidal boundary (BASELINE cfg 5), strips along x | cfg2_src: cfg2 + a "
                    "Coriolis field + Manning friction (the kernels with source terms)")
    ap.add_argument('--timing', action='store_true', help='-DSWE_WAVE_TIMING -DSWE_FLOW_TS_STAGE=S build of the library (THETIS_AMD_LIB): '
                    "per-block time stamps of stage S of the last flow launch, by the block's role")
    args = ap.parse_args()
    import torch
    import bench
    from thetis_amd import distributed

    class NoExchange(distributed.HaloExchanger):
        def start(self):
            return []

        def finish(self, reqs):
            pass

    class LoopbackP2P(object):
        """every peer's landing segment is my own: the push kernel stores into this device's zone, the wait finds the flags
        it raised itself (strips: what I send to a peer is as long as what I receive from it)"""
        def __init__(self, dev, part, rank, world, n_tracers=0, group=None, local_error=None):
            k = int(part.cells.shape[1])
            self.dev, self.n_channels = dev, 1
            dev.p2p_create([3*k] + [k]*int(n_tracers) + ([18] if k == 3 else []))
            self.n_channels = 1 + int(n_tracers) + (1 if k == 3 else 0)
            _, base, kind = dev.p2p_export()
            self.zone_kind = {1: 'uncached', 2: 'fine-grained', 3: 'device'}.get(kind, '?')
            peers = sorted(part.send)
            assert peers == sorted(part.recv) and all(part.send[q][1] == part.recv[q][1] for q in peers)
            dev.p2p_connect([base]*len(peers), [part.send[q][0] for q in peers], [part.send[q][1] for q in peers],
                            [part.recv[q][0] for q in peers], list(range(len(peers))), [len(part.recv_cells)]*len(peers),
                            n_from=len(peers))

        def timeouts(self):
            return self.dev.p2p_status(self.n_channels)[2]

    distributed.HaloExchanger = NoExchange
    distributed.P2PHalo = LoopbackP2P
    # one process plays one rank of `world`: the ranks' common decisions are this rank's own
    distributed.DistributedSwe2d._all_reduce = lambda self, values, op: np.asarray(list(values), dtype=float)
    distributed.DistributedSwe2d._all_reduce_int = lambda self, values: np.asarray(values, dtype=np.int64).ravel()
    distributed.DistributedSwe2d._ranks_share_a_device = lambda self: False
    kw = {}
    if args.case == 'cfg5':
        from thetis_amd import _lib
        from thetis_amd.mesh import RectangleMesh
        mesh = RectangleMesh(707, 354, 13800.0, 7200.0)
        bath = mesh.vertex_xy[:, 0]/2760.0
        uv, eta = np.zeros((mesh.num_cells, 3, 2)), np.zeros((mesh.num_cells, 3))
        dt = 0.1
    else:
        mesh, bath, uv, eta = bench.build_case(args.nx, args.nx//2) if args.nx else bench.build_case()
        dt = bench.DT*(1000.0/args.nx if args.nx else 1.0)
        if args.case in ('cfg4', 'cfg4_tracer_only'):
            kw = dict(n_tracers=1, use_limiter=True, tracer_only=(args.case == 'cfg4_tracer_only'))
    s = distributed.DistributedSwe2d(mesh, bath, dt, args.rank, args.world, 0, exchange_every=args.every, overlap_stages=args.overlap,
                                     exchange=('p2p' if args.exchange == 'p2p' else 'rccl'), split_last_stage=not args.nosplit,
                                     graph_mode=args.graph_mode,
                                     flow=(None if args.flow < 0 else bool(args.flow)), flow_exchange=(None if args.flowx < 0 else bool(args.flowx)), **kw)
    if args.case == 'cfg2_src':                   # the bench mesh with optional terms: Coriolis field + Manning friction (the SRC kernels)
        from thetis_amd import _lib
        s.dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        s.dev.set_field(_lib.FIELD_CORIOLIS, 1.0e-4)
        s.config_changed()
    if args.case == 'cfg5':
        s.dev.set_wetting_and_drying(0.4)
        s.dev.set_scalar(_lib.SCALAR_MANNING_DRAG, 0.02)
        if 2 in s.dev._marker_slot:
            s.dev.set_bc(2, {'elev': -0.5})
        s.config_changed()
    s.set_state_global(uv, eta)
    if kw:
        cxy = mesh.cell_xy()
        s.set_tracer_global(0, np.where(cxy[:, :, 0] < 40e3, 0.0, 30.0))
    p = s.part
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.prewarm:
        s.advance(args.every*25, use_graph=False)
        s.synchronize()
    s.set_state_global(uv, eta)
    s._capture(args.steps)
    best = 1e9
    for rep in range(3):
        s.set_state_global(uv, eta)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.advance(args.steps, use_graph=True)
        s.synchronize()
        best = min(best, time.perf_counter() - t0)
    to = s.p2p.timeouts() if s.p2p is not None else 0
    if args.timing:
        import ctypes
        fn = s.dev.lib.swe2d_debug_read_wave_timing
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        ts = np.zeros((6, 8192), dtype=np.uint64)
        s.dev._ck(fn(s.dev.h, ts.ctypes.data))
        nb = (p.num_cells + 63)//64
        t = (ts[:5, :nb] & np.uint64(0xffffffffff)).astype(np.int64)        # 40 bits of clock, as stored with the previous publish
        prev = (ts[5, :nb] & np.uint64(0xffffffffff)).astype(np.int64)
        # the blocks' roles, from the flow order the solver set (device ids -> flow position)
        from thetis_amd import ordering
        order = np.asarray(ordering.auto_cell_order(p, 0, p.num_cells))
        pos = np.empty(p.num_cells, dtype=np.int64)
        pos[order] = np.arange(p.num_cells)
        blk = pos//64
        role = np.zeros(nb, dtype=np.int64)                      # 1 ghost, 2 send, 4 boundary
        role[np.unique(blk[p.n_owned:])] |= 1
        role[np.unique(blk[np.asarray(p.send_cells)])] |= 2
        role[np.unique(blk[(np.asarray(p.cell_nbr) < 0).any(axis=1)])] |= 4
        ok = (t[0] > 0) & (prev > 0) & (t[4] >= t[0])
        us = lambda x: 0.01*x
        out = {}
        for name, sel in (('all', ok), ('interior', ok & (role == 0)), ('ghost', ok & ((role & 1) != 0)), ('send', ok & ((role & 2) != 0)),
                          ('boundary only', ok & (role == 4))):
            if not sel.any():
                continue
            q = lambda x: [round(float(v), 2) for v in np.percentile(us(x[sel]), [10, 50, 90, 100])]
            out[name] = {'blocks': int(sel.sum()), 'gap_prev_publish_to_stage_top': q(t[0] - prev), 'wait': q(t[1] - t[0]),
                         'arith': q(t[3] - t[2]), 'publish': q(t[4] - t[3]), 'period_prev_publish_to_publish': q(t[4] - prev)}
        print(json.dumps({'timing_p10_p50_p90_max_us': out}))
        dump = os.environ.get('RANKBENCH_TIMING_DUMP')
        if dump:                                                  # raw stamps (10 ns ticks) for offline analysis
            nbr = np.asarray(p.cell_nbr)
            rim = np.zeros(nb, dtype=np.int64)
            for f in range(nbr.shape[1]):
                okf = nbr[:, f] >= 0
                a, b = blk[np.nonzero(okf)[0]], blk[nbr[okf, f]]
                np.add.at(rim, a[a != b], 1)
            np.savez(dump, t=t, prev=prev, role=role, rim=rim, n_owned=p.n_owned, blk=blk)
    print(json.dumps({'case': args.case, 'exchange': args.exchange, 'split': not args.nosplit, 'p2p_timeouts': to,
                      'world': args.world, 'rank': args.rank, 'every': args.every, 'overlap': args.overlap, 'n_owned': int(p.n_owned),
                      'n_ghost': int(p.n_ghost), 'n_send': int(len(p.send_cells)), 'graph': s.graphed, 'graph_mode': s.graph_mode, 'flow': bool(s.flow), 'flow_exchange': bool(s.flow_exchange), 'flow_timeouts': s.dev.flow_timeouts(), 'fused_pair': list(s.dev.fused_pair_info()) if hasattr(s.dev, 'fused_pair_info') else None,
                      'us_per_step': 1e6*best/args.steps}))


if __name__ == '__main__':
    main()
