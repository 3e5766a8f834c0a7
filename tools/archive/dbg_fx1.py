"""debug (round 6): the dataflow launches with the exchange inside, one exchange per step, four ranks sharing a GPU - the candidate first contact
dropped.  python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29655 tools/archive/dbg_fx1.py [order]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from bench import build_case, DT
from thetis_amd.distributed import DistributedSwe2d
from thetis_amd.device import Swe2dDevice
from thetis_amd.partition import strip_owner, build_partition

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group(backend='gloo', rank=rank, world_size=world)
ctrl = dist.new_group(backend='gloo')
mesh, bath, uv, eta = build_case(256, 64)
owner = strip_owner(mesh, world)
order = sys.argv[1] if len(sys.argv) > 1 else '2k1'        # e.g. "1", "2k1" (keep the every-2 solver alive), "2c1" (close it first), "h2k1"
n_check = int(os.environ.get('N_CHECK', '9'))
one = Swe2dDevice(mesh, bath, DT, device_id=0)
one.set_state(uv, eta)
one.advance(n_check)
u1, e1 = one.get_state()
one.close()
parts, alive = {}, []


def run(every, exchange='p2p', flow=True, split=False, mode='cycle', pre=0):
    if every not in parts:
        parts[every] = build_partition(mesh, owner, rank, halo_depth=3*every)
    s = DistributedSwe2d(mesh, bath, DT, rank, world, 0, exchange_every=every, overlap_stages=0, graph_mode=mode, exchange=exchange,
                         split_last_stage=split, partition=parts[every], group=ctrl, flow=flow)
    if flow:
        assert bool(s.flow)
    s.set_state_global(uv, eta)
    if pre:
        s.advance(pre, use_graph=False)
        s.synchronize()
        s.set_state_global(uv, eta)
    s.advance(n_check, use_graph=False)
    s.synchronize()
    ids, u, e = s.get_state_owned()
    bad = np.nonzero((u != u1[ids]).any(axis=(1, 2)) | (e != e1[ids]).any(axis=1))[0]
    col = (ids[bad]//2) % 256
    print('rank {:d} every {:d} exchange {:} flow {:} flowx {:}: {:d} of {:d} owned cells differ; columns {:}; max |du| {:.3e}; timeouts {:}'.format(
        rank, every, exchange, flow, bool(getattr(s, 'flow_exchange', False)), len(bad), len(ids), sorted(set(col.tolist()))[:20],
        float(np.abs(u - u1[ids]).max()), s.p2p.timeouts() if s.p2p is not None else None), flush=True)
    return s


tok = order.replace('k', ' k ').replace('c', ' c ').split()
for t in tok:
    if t == 'k':
        continue
    if t == 'c':
        for s in alive:
            s.close()
        alive = []
        continue
    if t.startswith('h'):
        alive.append(run(int(t[1:]), exchange='host', flow=False, split=True, mode='none'))
    elif t.startswith('s'):                 # stage launches over p2p
        alive.append(run(int(t[1:]), flow=False))
    elif t.startswith('p'):                 # with a long run before (the soak)
        alive.append(run(int(t[1:]), pre=2400))
    else:
        alive.append(run(int(t)))
    dist.barrier(group=ctrl)
for s in alive:
    s.close()
dist.destroy_process_group()
