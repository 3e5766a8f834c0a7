#!/usr/bin/env python
"""First contact with a multi-GPU node, readable: ``python -m tools.first_contact --gpus 8``.

Starts ``bench.py --gpus N`` under ``torch.distributed.run`` exactly as the driver does (one rank per GPU, 127.0.0.1 rendez-vous)
and tells the story of the run instead of one JSON line: which transports set up and reproduced the host-staged exchange, what the
soak found, every schedule candidate with its time, what was chosen and WHY (the fastest verified candidate inside the set-up
budget), the timed region, and every failure with the step it happened in.  The running log of the ranks (tools/benchlib.py
``progress``: rank 0, stderr, flushed line by line) is passed through as it comes, so a run that dies half way leaves everything up
to that point on the terminal and in ``--log``.  Nothing here is on the product path: it wraps the bench.

    python -m tools.first_contact --gpus 8                    # the BASELINE cfg 3 mesh, 200 timed steps
    python -m tools.first_contact --gpus 2 --same-gpu         # test boxes: the ranks share the visible GPU (gloo instead of RCCL)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def summarise(d):
    """the lines a reader of a SCALE log needs, from the bench's JSON line"""
    cfg = d.get('config', {})
    out = []
    out.append('result: {:.4f} ms per step, {:.3e} element-updates/s on {:d} GPUs ({:})'.format(
        d.get('ms_per_step') or float('nan'), d.get('value') or 0.0, d.get('n_gpus', 0), cfg.get('workload', '?')))
    out.append('transports verified bit for bit against the host-staged exchange: {:}'.format(cfg.get('transports_verified')))
    soak = cfg.get('soak') or {}
    for v in soak.get('verified', []):
        out.append('  soak: {:} - {:d} steps in {:.1f} s ended on the single-device bits'.format(v['what'], v['steps'], v['seconds']))
    tuning = sorted(cfg.get('schedule_tuning') or [], key=lambda t: t['us_per_step'])
    if tuning:
        out.append('schedule candidates (us per step, max over ranks; fastest first):')
        for t in tuning:
            out.append('  {:7.2f}  {:4s} every {:d}  overlap {:d}  split {:}  graphs {:5s}  dataflow {:}'.format(
                t['us_per_step'], t['exchange'], t['exchange_every'], t['overlap_stages'], t['split_last_stage'], t['graph_mode'], t['flow']))
    out.append('chosen: exchange {:} ({:}), every {:} steps, dataflow launches {:}, exchange inside the launch {:}, graphs {:} - the '
               'fastest candidate that every rank verified inside the set-up budget ({:.0f} of {:.0f} s used)'.format(
                   cfg.get('exchange'), cfg.get('exchange_transport'), cfg.get('exchange_every'), cfg.get('flow'), cfg.get('flow_exchange'),
                   cfg.get('graph_mode'), cfg.get('setup_s') or 0.0, cfg.get('setup_budget_s') or 0.0))
    if cfg.get('setup_skipped'):
        out.append('not tried (budget): {:}'.format(cfg['setup_skipped']))
    out.append('volume conserved: {:}; p2p time-outs: {:}; flow time-outs: {:}; exchange share of a step: {:}'.format(
        cfg.get('volume_conserved'), cfg.get('p2p_timeouts'), cfg.get('flow_timeouts'), cfg.get('exchange_time_fraction')))
    lm = cfg.get('large_mesh')
    if lm:
        out.append('8x mesh ({:}): {:.4f} ms per step, {:.2f} of the HBM roofline per GPU, volume conserved {:}'.format(
            lm.get('workload'), lm.get('ms_per_step'), lm.get('frac_of_hbm_roofline_per_gpu'), lm.get('volume_conserved')))
    fails = cfg.get('failures') or []
    out.append('failures: {:}'.format('none' if not fails else ''))
    for f in fails:
        out.append('  - ' + f)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=8)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--port', type=int, default=29533)
    ap.add_argument('--same-gpu', action='store_true', help='ranks share the visible GPU(s): gloo control plane only, no RCCL (test boxes)')
    ap.add_argument('--mesh', default='', help='"nx,ny": another channel mesh than BASELINE cfg 3 (test boxes)')
    ap.add_argument('--log', default='', help='also write the running log and the summary to this file')
    args = ap.parse_args()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.same_gpu:
        env['THETIS_AMD_DIST_BACKEND'] = 'gloo'
    if args.mesh:
        env['THETIS_AMD_BENCH_MESH'] = args.mesh
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(args.port), os.path.join(ROOT, 'bench.py'), '--gpus', str(args.gpus), '--steps', str(args.steps),
           '--warmup', str(args.warmup)]
    log = open(args.log, 'w') if args.log else None

    def say(line):
        print(line, flush=True)
        if log:
            log.write(line + '\n')
            log.flush()
    say('$ ' + ' '.join(cmd))
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1)
    last_json = None
    for line in p.stdout:
        line = line.rstrip('\n')
        if line.startswith('{') and '"metric"' in line:
            last_json = line
            continue
        say(line)
    rc = p.wait()
    say('--- torch.distributed.run exited with {:d}'.format(rc))
    if last_json is None:
        say('NO JSON LINE: the run died before rank 0 printed it; the log above ends where it stopped')
        return 1
    for line in summarise(json.loads(last_json)):
        say(line)
    say(last_json)
    return 0 if rc == 0 else rc


if __name__ == '__main__':
    sys.exit(main())
