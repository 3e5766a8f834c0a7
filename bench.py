#!/usr/bin/env python
"""
bench.py - headline benchmark of the hot path (BASELINE.json: "DG element-updates/sec + achieved HBM GB/s,
2D SWE SSPRK33 at 1/2/4/8 GPUs").

Workload = BASELINE.json configs[1]/[2] (SURVEY.md 8d cfg 2/3): RectangleMesh(1000, 500, 100e3, 50e3) = 1,000,000
triangles, DG-P1, flat bathymetry h=20, eta0 = 0.5 exp(-r^2/(5km)^2) + U(-1e-3,1e-3) noise, u0 noise +-1e-3
(numpy default_rng(1234)), closed walls, sigma_LF=1, dt=0.25 s.  One "step" = one SSPRK33 time step = 3 stage
kernels = 3 element-updates per cell.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU;
                                                        the same 1M-triangle mesh is strip-partitioned: strong scaling)

Clock pre-warm: an MI355X needs ~0.2-0.3 s of sustained load to settle its clocks (measured: 0.147 ms/step right after start
with W = 10, 0.128 ms/step after 0.3 s of the same work), and W warm-up steps of 0.14 ms are over long before that.  Since
production runs are thousands of steps, the bench first runs PREWARM_S seconds of the same time stepping (i.e. extra
untimed warm-up steps; the state simply keeps evolving - a host-side state reset would idle the GPU for ~20 ms and cool it
down again) and only then does the W untimed warm-up steps and the K timed steps; `config.prewarm_s` reports it
(--prewarm 0 switches it off).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NX, NY, LX, LY = 1000, 500, 100e3, 50e3
DT = 0.25
PREWARM_S = 0.5                          # seconds of untimed stepping before the warm-up (clock settling), see above
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_ELEMENT_STEP = 684.0          # 180 + 252 + 252 algorithmic bytes per triangle per step
BYTES_PER_ELEMENT_UPDATE = BYTES_PER_ELEMENT_STEP/3.0
# what the FUSED algorithm must move per triangle and step in the same perfect-cache model: stages 1 + 2 in one
# launch read U(0) and the static data once and write U(2) - 72 + 36 + 72 = 180 B, U(1) never leaves the chip -, stage 3 its 252 B
FUSED_BYTES_PER_ELEMENT_STEP = 180.0 + 252.0
# ... and with all three stages in one launch: U(0) and the static data read once, U(3) written
TRIPLE_BYTES_PER_ELEMENT_STEP = 180.0
TRAFFIC_JSON = 'r06_traffic.json'           # committed PMC passes of this library's kernels on this workload (profiles/README.md)
TRAFFIC_4M_JSON = 'r06_traffic_4m.json'     # ... on the 4M-triangle mesh behind roofline.beyond_cache
FP64_CLOCK_HZ = 2.4e9                       # MI355X_MICROARCH.md: max clock 2400 MHz
N_SIMD = 1024                               # 256 CUs x 4 SIMDs; a wave64 FP64-rate instruction issues in 4 cycles (16 lanes per clock)
BEYOND_CACHE_NX, BEYOND_CACHE_NY = 2000, 1000   # 4M triangles: 3 x 288 MB of state, beyond the 256 MB Infinity Cache


def build_case(nx=NX, ny=NY):
    from thetis_amd.mesh import RectangleMesh
    mesh = RectangleMesh(nx, ny, LX, LY)
    bath = np.full(mesh.num_vertices, 20.0)
    n = mesh.num_cells
    rng = np.random.default_rng(1234)
    cxy = mesh.cell_xy()
    eta = 0.5*np.exp(-((cxy[:, :, 0] - 50e3)**2 + (cxy[:, :, 1] - 25e3)**2)/(5e3)**2)
    eta = eta + 1e-3*rng.uniform(-1, 1, size=(n, 3))
    uv = 1e-3*rng.uniform(-1, 1, size=(n, 3, 2))
    return mesh, bath, uv, eta


def cpu_baseline(budget_s=10.0):
    """The oracle's C restatement (oracle/swe2d_ref.c, swe2d_ref_advance_blocked: persistent OpenMP team, cell blocks owned and
    first touched by one thread each, fused stage update) timed on the host cores on a bounded number of steps of the same
    1M-triangle workload - in a process of its own (oracle/cpu_bench.py) so that the OpenMP runtime starts with thread
    binding; two placements are tried (one thread per hardware thread, one per second hardware thread) and the faster one is
    reported.  Returns (cpu_baseline dict, steps, uv, eta of the reported run)."""
    import subprocess
    import tempfile
    best, best_state = None, None
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    # a container may see every hardware thread of the host and still be held to a CPU-time quota (cgroup v2 cpu.max, e.g.
    # "1600000 100000" = 16 CPUs on the 256-thread GPU boxes): more threads than that only contend for the same 16 CPUs
    quota = ncpu
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = max(1, min(ncpu, int(float(q)/float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    tried = []
    placements = [('one thread per CPU of the cgroup quota, spread over the cores', quota, 'spread')]
    if quota < ncpu:
        placements.append(('twice the cgroup quota', min(ncpu, 2*quota), 'spread'))
    else:
        placements.append(('every second hardware thread', max(1, ncpu//2), 'spread'))
    with tempfile.TemporaryDirectory() as tmp:
        for label, nthreads, bind in placements:
            env = dict(os.environ)
            env.update({'OMP_PROC_BIND': bind, 'OMP_PLACES': 'cores', 'OMP_NUM_THREADS': str(nthreads)})
            out_npz = os.path.join(tmp, 'state{:d}.npz'.format(nthreads))
            try:
                r = subprocess.run([sys.executable, '-m', 'oracle.cpu_bench', '--nx', str(NX), '--ny', str(NY), '--budget',
                                    str(budget_s/2.0), '--out', out_npz], capture_output=True, text=True, env=env, cwd=ROOT,
                                   timeout=600)
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
            except Exception as e:                       # the baseline is reported, never fatal
                tried.append({'placement': label, 'error': str(e)[:200]})
                continue
            tried.append({'placement': label, 'threads': d['cores'], 'value': d['value'], 'speedup_over_1core': d['speedup_over_1core']})
            if best is None or d['value'] > best['value']:
                best, best_label = d, label
                z = np.load(out_npz)
                best_state = (int(z['steps']), z['uv'], z['eta'])
            if quota == 1:
                break
    if best is None:
        return {'value': None, 'unit': 'element-updates/s', 'cores': 0, 'kind': 'port', 'sample': 'failed', 'tried': tried}, None
    out = {'value': best['value'], 'unit': 'element-updates/s', 'cores': best['cores'], 'kind': 'port',
           'sample': 'median of 3 runs of {:d} SSPRK33 steps of the same 1M-triangle workload, oracle/swe2d_ref.c '
                     '(swe2d_ref_advance_blocked: one OpenMP region, static cell blocks first-touched by their owner, fused '
                     'stage update), {:d} threads ({:}); the host shows {:d} hardware threads, the cgroup CPU quota is {:d}; '
                     'OMP_PROC_BIND={:} OMP_PLACES=cores, {:.1f} s per run'.format(
                         best['steps'], best['cores'], best_label, ncpu, quota, best['omp']['OMP_PROC_BIND'],
                         float(np.median(best['seconds']))),
           'cpu_quota': quota, 'hardware_threads': ncpu,
           'value_1core': best['value_1core'], 'speedup_over_1core': best['speedup_over_1core'],
           'sample_1core': 'median of 3 runs of 2 steps on one thread', 'placements': tried}
    return out, best_state


def launch_structure(dev):
    """'triple' (all three stages of a step in one launch), 'pair' (stages 1 + 2 fused, stage 3 a stage launch) or 'stages'"""
    if dev.fused_triple_info()[0]:
        return 'triple'
    return 'pair' if dev.fused_pair_info()[0] else 'stages'


MODEL_BYTES = {'stages': BYTES_PER_ELEMENT_STEP, 'pair': FUSED_BYTES_PER_ELEMENT_STEP, 'triple': TRIPLE_BYTES_PER_ELEMENT_STEP}
LAUNCHES = {'stages': 3, 'pair': 2, 'triple': 1}


def measured_traffic(n_cells, structure, name=None, want_cells=1000000):
    """HBM bytes per element-update (a third of a step) and VALU wave-instructions per step from the committed PMC passes
    (rocprofv3 cannot run inside the timed bench): FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU collected in separate --pmc runs, the
    fetch counter corrected with the calibration copy kernel, see profiles/README.md.  Only valid for the workload AND the launch
    structure it was measured on: with the stages fused the file must hold the fused kernel, without it must not - else nothing
    is reported (a static figure next to a run it does not belong to)."""
    name = name or TRAFFIC_JSON
    path = os.path.join(ROOT, 'profiles', name)
    try:
        with open(path) as f:
            t = json.load(f)
        in_file = 'triple' if 'fused_stage_triple_kernel' in t else ('pair' if 'fused_stage_pair_kernel' in t else 'stages')
        if n_cells == want_cells and structure == in_file:
            return float(t['traffic_bytes_per_launch']), t.get('valu_wave_instructions_per_step'), 'profiles/' + name
    except (OSError, KeyError, ValueError):
        pass
    return None, None, None


def beyond_cache(args):
    """The same stage kernel on a 4M-triangle mesh of the same channel (state 3 x 288 MB: nothing of it stays in the 256 MB
    Infinity Cache from one stage to the next, which the 1M-triangle headline workload - 3 x 72 MB - largely does):
    roofline.frac_beyond_cache, measured like roofline.frac (HIP events around K back-to-back steps on the launch stream)."""
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = build_case(BEYOND_CACHE_NX, BEYOND_CACHE_NY)
    n = mesh.num_cells
    dev = Swe2dDevice(mesh, bath, DT/2.0, device_id=0)
    dev.set_state(uv, eta)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < max(0.3, args.prewarm):
        dev.advance(50)
        dev.synchronize()
    steps = max(10, min(50, args.steps))
    ms_events = min(dev.advance_timed(steps, per_launch=False)[0] for _ in range(2))
    ms_kernel = ms_events/(3.0*steps)
    assert np.isfinite(dev.diagnostics()).all()
    structure = launch_structure(dev)
    dev.close()
    achieved = BYTES_PER_ELEMENT_UPDATE*n/(ms_kernel*1e-3)/1e9
    # committed PMC passes of THIS library's launches on this very workload (profiles/README.md)
    traffic, valu, src = measured_traffic(n, structure, TRAFFIC_4M_JSON, 4000000)
    model = MODEL_BYTES[structure]*n
    return {'frac_beyond_cache': achieved/HBM_PEAK_GBS,
            'beyond_cache': {'workload': 'RectangleMesh({:d},{:d}) = {:d} triangles, same channel and kernels'.format(
                                 BEYOND_CACHE_NX, BEYOND_CACHE_NY, n),
                             'achieved': achieved, 'avg_launch_ms': ms_kernel, 'steps': steps, 'traffic': traffic,
                             'traffic_source': src, 'launch_structure': structure, 'launches_per_step': LAUNCHES[structure],
                             'frac_fused_model': model/(ms_events/steps*1e-3)/1e9/HBM_PEAK_GBS,
                             'traffic_rate_frac': (3.0*traffic/(ms_events/steps*1e-3)/1e9/HBM_PEAK_GBS) if traffic else None,
                             'valu_issue_frac': (valu*4.0/(N_SIMD*FP64_CLOCK_HZ)/(ms_events/steps*1e-3)) if valu else None,
                             'algorithmic_bytes_per_launch': BYTES_PER_ELEMENT_UPDATE*n,
                             'element_updates_per_s': n*3.0*steps/(ms_events*1e-3)}}


def run_single(args):
    from thetis_amd.device import Swe2dDevice
    mesh, bath, uv, eta = build_case()
    n = mesh.num_cells
    dev = Swe2dDevice(mesh, bath, DT, device_id=0)
    dev.set_state(uv, eta)
    if args.prewarm > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.prewarm:
            dev.advance(200)
            dev.synchronize()
    dev.advance(args.warmup)
    dev.synchronize()
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # the K timed steps, bracketed on the launch stream by two HIP events (swe2d_advance_timed) and by the host clock
    ms_events, _ = dev.advance_timed(args.steps, per_launch=False)
    dev.synchronize()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    # mean stage-kernel launch duration over the timed region: 3 back-to-back launches per step, no gaps in between
    ms_kernel = ms_events/(3.0*args.steps)
    # the spread of the figure: the same K-step region four more times (the state keeps evolving, no host reset in between);
    # `value`, `ms_per_step` and `roofline.frac` stay those of the FIRST region (the contract's timed K steps), the samples say
    # how far one 2-3 ms region is from the next on this box (clock state, Infinity Cache contents)
    ms_samples = [ms_events] + [dev.advance_timed(args.steps, per_launch=False)[0] for _ in range(4)]
    frac_samples = [BYTES_PER_ELEMENT_UPDATE*n/(ms/(3.0*args.steps)*1e-3)/1e9/HBM_PEAK_GBS for ms in ms_samples]
    # cross-check (separate pass): events around every single launch; ~3 % higher because of the event brackets
    n_ev = min(args.steps, 50)
    _, ms_kernel_each = dev.advance_timed(n_ev, per_launch=True)
    d = dev.diagnostics()
    assert np.isfinite(d).all()
    value = n*3.0*args.steps/t_wall
    achieved = BYTES_PER_ELEMENT_UPDATE*n/(ms_kernel*1e-3)/1e9
    fused = dev.fused_pair_info()
    structure = launch_structure(dev)
    traffic, valu, traffic_src = measured_traffic(n, structure)
    step_s = ms_events/args.steps*1e-3
    model_step = MODEL_BYTES[structure]*n
    out = {
        'metric': 'DG element-updates/sec, 2D SWE DG-P1 SSPRK33',
        'value': value, 'unit': 'element-updates/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3*t_wall/args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'BASELINE cfg2: RectangleMesh(1000,500,100e3,50e3) = 1M triangles, DG-P1 SWE, SSPRK33, '
                               'flat h=20, closed walls, dt=0.25', 'n_cells': n, 'parallelism': 'single',
                   'prewarm_s': args.prewarm},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     # `frac` keeps SURVEY.md 8d's definition: the bytes of THREE stage launches (228 B per element-update) over the
                     # time - a useful-work rate.  With stages fused the launches move fewer bytes than that model, so `frac` contains an
                     # algorithmic saving and could pass 1 as fusion deepens; the three figures after it say what the hardware did:
                     'frac': achieved/HBM_PEAK_GBS, 'frac_samples': frac_samples, 'frac_median': float(np.median(frac_samples)),
                     'frac_min': float(min(frac_samples)), 'frac_max': float(max(frac_samples)),
                     'frac_definition': 'SURVEY 8d model bytes of three stage launches (684 B per triangle and step) / step time / 8 TB/s',
                     # (1) the bytes the launch structure that RAN must move in the same perfect-cache model (all three stages in one
                     #     launch: 180 B per triangle and step; fused pair + stage 3: 432; three stage launches: 684 = frac) over the time
                     'frac_fused_model': model_step/step_s/1e9/HBM_PEAK_GBS,
                     'fused_model_bytes_per_step': model_step,
                     # (2) what the memory system delivered by the counters (profiles/, same library and workload)
                     'traffic': traffic, 'traffic_unit': 'bytes per element-update = a third of a step (three stage launches: per launch)',
                     'traffic_rate_GBs': (traffic/(ms_kernel*1e-3)/1e9) if traffic else None,
                     'traffic_rate_frac': (traffic/(ms_kernel*1e-3)/1e9/HBM_PEAK_GBS) if traffic else None,
                     'traffic_source': traffic_src,
                     # (3) how busy the FP64 pipes were: VALU wave-instructions per step by the counters x 4 issue cycles over
                     #     1024 SIMDs x 2.4 GHz, against the step time (a kernel near 1 here is bound by its arithmetic, not by HBM)
                     'valu_wave_instructions_per_step': valu,
                     'valu_issue_frac': (valu*4.0/(N_SIMD*FP64_CLOCK_HZ)/step_s) if valu else None,
                     'kernel': {'triple': 'swe_fuse123_kernel (all three stages of a step in one launch on two-ring tiles, U(1) and U(2) in LDS): '
                                          'avg_launch_ms is per element-update = a third of the launch',
                                'pair': 'swe_fuse12_kernel (stages 1 + 2 in one launch) + swe_stage_kernel (stage 3): avg_launch_ms is per '
                                        'element-update = a third of a step',
                                'stages': 'swe_stage_kernel'}[structure],
                     'launch_structure': structure, 'launches_per_step': LAUNCHES[structure],
                     'fused_stage_pair': {'tiles': fused[1], 'ring_cells': fused[2]} if (fused[0] and structure == 'pair') else None,
                     'fused_stage_triple': (dict(zip(('tiles', 'ring1_cells', 'ring2_cells'), dev.fused_triple_info()[1:]))
                                            if structure == 'triple' else None),
                     'avg_launch_ms': ms_kernel, 'avg_launch_ms_per_launch_events': ms_kernel_each,
                     # the mean duration of ONE launch by the per-launch events: what the launch-weighted mean of the kernel rows of
                     # `rocprofv3 --kernel-trace --stats` of this command comes to (profiles/)
                     'avg_kernel_launch_ms': ms_kernel_each*3.0/LAUNCHES[structure],
                     'algorithmic_bytes_per_launch': BYTES_PER_ELEMENT_UPDATE*n},
    }
    if not args.no_beyond_cache:
        out['roofline'].update(beyond_cache(args))
    if not args.no_cpu:
        cb, state = cpu_baseline()
        out['cpu_baseline'] = cb
        if state is not None:
            # parity of the two paths on the baseline's sample (reported)
            steps_c, u_c, e_c = state
            dev.set_state(uv, eta)
            dev.advance(steps_c)
            u_g, e_g = dev.get_state()
            out['cpu_baseline']['gpu_vs_cpu_rel_linf'] = {
                'uv': float(np.abs(u_g - u_c).max()/np.abs(u_c).max()),
                'eta': float(np.abs(e_g - e_c).max()/np.abs(e_c).max()), 'steps': steps_c}
    dev.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-beyond-cache', action='store_true', help='skip the 4M-triangle run behind roofline.frac_beyond_cache')
    ap.add_argument('--prewarm', type=float, default=PREWARM_S, help='seconds of untimed stepping before the warm-up steps')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 or world > 1 or os.environ.get('THETIS_AMD_FORCE_DIST'):   # env: exercise the N>1 code path on one GPU
        from tools.benchlib import run_distributed_bench
        case = build_case
        if os.environ.get('THETIS_AMD_BENCH_MESH'):       # tests: "nx,ny" - many ranks sharing the one GPU of a test box
            nx, ny = (int(v) for v in os.environ['THETIS_AMD_BENCH_MESH'].split(','))
            case = lambda: build_case(nx, ny)
        run_distributed_bench(args, case, DT, BYTES_PER_ELEMENT_UPDATE, HBM_PEAK_GBS)
    else:
        run_single(args)


if __name__ == '__main__':
    main()
