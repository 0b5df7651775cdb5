# -*- coding: utf-8 -*-
'''
bench.py -- headline benchmark of the hot path (BASELINE.json):
  IPCS (rotational pressure-correction) time-steps/s on the Karman-vortex-street
  channel, P2-P1 Taylor-Hood, ~10 M DoF, plus the pressure-Poisson SpMV GB/s
  against the MI355X HBM roofline.

A "step" is one pass of the reference driver's loop body
(tests/test_karman_vortex_street.py:219-286 of the reference): Rotational.step()
(tentative velocity -> pressure Poisson -> velocity correction) followed by the
CFL step-size controller, on synthetic data (structured channel mesh pulled
onto the cylinder: body-fitted, flow_amd/fem/mesh.py).

  python bench.py --gpus N --steps K --warmup W

Everything is measured in mode 'parity' (flow_amd.navier_stokes: the
reference's Newton path -- start from u0, exact Newton steps -- which
reproduces the reference's iterate to < 1e-6, tests/test_full_size_parity.py).

The headline `value` (round 6: the metric names a vortex street) is the mean
over ONE SHEDDING PERIOD OF THE DEVELOPED STREET: 2400 untimed steps behind the
settled plateau (setup), W + K steps (the window the flags ask for, reported as
`value_developed`: it happens to sit in one of the bursts of the run), then
`--developed-period` (400) steps timed as a whole between two barriers --
`value`, `ms_per_step`, `headline_steps`.  The K-step window on the early
plateau of the run (symmetric flow, one Newton iteration per step: rounds 1-5's
`value`) is `value_plateau`.  At N = 1 the plateau window is also repeated in
mode 'fast' (`config.fast_mode`) and with every start vector off
(`config.zero_start`).  `--headline plateau`, or a run without the developed
phase (`--developed 0`, sizes other than the headline's unless asked for), has
`value` = the plateau window and says so in `config.headline`.

N > 1 (launched by torch.distributed.run, one rank per GPU): strong scaling,
the mesh is fixed (flow_amd/parallel.py).

Prints ONE JSON line on rank 0.
'''
from __future__ import print_function

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def spmv_bytes(n, nnz):
    '''Algorithmic bytes of one CSR SpMV.'''
    return 12 * nnz + 4 * (n + 1) + 8 * n + 8 * n


def measure_spmv_replay(apply, n, reps=100, warmup=10):
    '''Average launch duration (s) of back-to-back launches of an SpMV, timed
    with HIP events on the stream the kernel is launched on (torch's current
    stream is the library's stream: flow_amd/device.py).'''
    import torch
    from flow_amd import device
    x = torch.sin(torch.arange(n, dtype=torch.float64, device=device.get()))
    y = device.empty(n)
    for _ in range(warmup):
        apply(x, y)
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        apply(x, y)
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) * 1.0e-3 / reps


def measure_stream_ceilings(nbytes=1 << 30, reps=20):
    '''What plain streaming kernels sustain on this box: own
    16-byte-per-lane kernels (flow_profile_stream_copy / _read), `nbytes` read
    (+ as many written for the copy) per launch -- far beyond the 256 MB
    Infinity Cache --, HIP events on the launch stream.  Returns GB/s of the
    bytes moved: (copy, read-only).'''
    import torch
    from flow_amd import device, _hip
    n = nbytes // 8
    src = torch.ones(n, dtype=torch.float64, device=device.get())
    dst = device.empty(n)
    lib = _hip.lib()

    def timed(go):
        for _ in range(3):
            go()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            go()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1.0e-3 / reps
    t_copy = timed(lambda: _hip.check(lib.flow_profile_stream_copy(
        n, _hip.f64(src), _hip.f64(dst), _hip.stream())))
    t_read = timed(lambda: _hip.check(lib.flow_profile_stream_read(
        n, _hip.f64(src), _hip.f64(dst), _hip.stream())))
    del src, dst
    return 2.0 * nbytes / t_copy / 1e9, nbytes / t_read / 1e9


def measure_spmv_in_solver(prob, rows, steps, tol):
    '''The roofline kernel as the pressure CG runs it: `steps` more time steps
    with every launch of the fused-dot SpMV of the `rows`-row operator
    bracketed by HIP events on the launch stream (flow_profile_spmv_begin /
    _end of the C ABI) -- the matrix competes for the caches with the V-cycle's
    transfer operators, exactly as in the timed region.  Returns (average
    seconds per launch, launches).'''
    from flow_amd import _hip
    lib = _hip.lib()
    _hip.check(lib.flow_profile_spmv_begin(int(rows), 4096))
    try:
        for _ in range(steps):
            prob.step(tol=tol)
    finally:
        total = ctypes.c_double(0.0)
        count = ctypes.c_int(0)
        _hip.check(lib.flow_profile_spmv_end(ctypes.byref(total),
                                             ctypes.byref(count)))
    # an event pair also reads the dispatch latency of the launch it brackets
    # (a profiler's kernel duration does not): what the same pair reads
    # around a null kernel is reported beside it, NOT taken out -- the figure
    # errs on the slow side (rocprofv3 sees the same launches ~3 us shorter)
    over = ctypes.c_double(0.0)
    _hip.check(lib.flow_profile_event_overhead(ctypes.byref(over),
                                               _hip.stream()))
    raw = total.value * 1.0e-6 / max(count.value, 1)
    return raw, count.value, raw, over.value * 1.0e-6


def measure_spmv_hbm_resident(rows=10000000, band=1540, reps=50):
    '''The same kernel on a 7-point banded matrix of 1.04 GB, which cannot sit
    in the 256 MB Infinity Cache.'''
    import numpy
    import scipy.sparse as sp
    from flow_amd.fem.multigrid import CsrOperator
    from flow_amd import _hip
    offs = numpy.array([-band - 1, -band, -1, 0, 1, band, band + 1])
    i = numpy.arange(rows)[:, None] + offs[None, :]
    ok = (i >= 0) & (i < rows)
    rowptr = numpy.concatenate([[0], numpy.cumsum(ok.sum(axis=1))])
    vals = numpy.where(numpy.broadcast_to(offs, i.shape)[ok] == 0, 6.0, -1.0)
    A = sp.csr_matrix((vals, i[ok], rowptr), shape=(rows, rows))
    del i, ok, vals
    op = CsrOperator(A)
    lib = _hip.lib()

    def apply(x, y):
        _hip.check(lib.flow_operator_apply(
            ctypes.byref(op.op), _hip.f64(x, rows), _hip.f64(y, rows),
            _hip.stream()))
    t = measure_spmv_replay(apply, rows, reps=reps, warmup=5)
    nbytes = spmv_bytes(rows, A.nnz)
    return {'rows': rows, 'bytes_per_launch': nbytes,
            'us_per_launch': t * 1e6, 'GBps': nbytes / t / 1e9,
            'frac': nbytes / t / 1e9 / HBM_PEAK_GBPS}


def cpu_baseline(Kbc, isbc, b, tol, gpu, ndofs, budget_s=12.0):
    '''The pressure solve of this workload on the box's host cores with the
    SAME algorithm the GPU runs: CG preconditioned with the smoothed-
    aggregation V(1,1) cycle on the same hierarchy (oracle/cpu_cg.c, OpenMP,
    first-touched copies), same right-hand side, same stopping test.  Beside
    it, like for like: Jacobi-CG iterations/s and SpMV GB/s on both sides, and
    the full CPU-oracle step at the sizes it finishes.'''
    import numpy
    from oracle import cpu_lib
    from flow_amd.fem.multigrid import Multigrid
    try:
        lib = cpu_lib.load(cpu_lib.build(native=True, out_dir='/tmp'))
    except Exception:                                  # noqa: BLE001
        lib = cpu_lib.load()
    cores = lib.oracle_num_threads()
    n = Kbc.layout.N
    mg = Multigrid(Kbc, isbc, keep_host=True)
    hier = cpu_lib.MgHierarchy(lib, mg.host_levels, mg.Ainv_host, mg.omega)
    hier.cg(b, tol, maxit=2)                               # page in / warm up
    t0 = time.perf_counter()
    x_cpu, its, _res, ok = hier.cg(b, tol, maxit=1000)
    t_first = time.perf_counter() - t0
    solves = int(max(1, min(50, budget_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(solves):
        hier.cg(b, tol, maxit=1000)
    t_solve = (time.perf_counter() - t0) / solves
    # Jacobi-CG and SpMV, same matrix
    A = mg.host_levels[0][0]
    Ac = cpu_lib.Csr(lib, A)
    xv = cpu_lib.Vec(lib, numpy.sin(numpy.arange(n, dtype=float)))
    yv = cpu_lib.Vec(lib, numpy.zeros(n))
    for _ in range(3):
        lib.oracle_csr_spmv(Ac.h, xv.h, yv.h)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.oracle_csr_spmv(Ac.h, xv.h, yv.h)
    spmv_s = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    _, jits, _, _ = cpu_lib.jacobi_cg(lib, A, b, 1e-30, maxit=200)
    jac_rate = jits / (time.perf_counter() - t0)
    steps, fit = oracle_step_timing(ndofs)
    big = steps[-1]
    try:
        iterative = cpu_iterative_step()
    except Exception as exc:                           # noqa: BLE001
        iterative = {'error': repr(exc)}
    out = {
        # The whole step on the CPU, MEASURED on a bounded sample of this
        # workload and scaled to the metric's unit LINEARLY with the DoF count
        # (which flatters the CPU).  Round 6: the sample is the step with
        # ITERATIVE solvers on all host cores where its pieces are threaded
        # (oracle/cpu_step.py; `iterative_step` below has the phases and which
        # of them ran on how many cores); the one-core sparse-LU oracle step
        # (rounds 1-5's value) keeps its own keys: `oracle_step_scaled`,
        # `oracle_step*`.  Baseline only.
        'value': big['dofs_per_s'] / float(ndofs),
        'unit': 'time-steps/s',
        'cores': 1,
        'kind': 'port',
        'sample': 'oracle step() (oracle/fem_oracle.py: one Rotational step, '
                  'sparse LU in every Newton iteration and for both linear '
                  'systems, one core) MEASURED on %s (%d DoF: %.2f s per '
                  'step = %.0f DoF/s), scaled linearly with the DoF count to '
                  'the %d DoF of this workload (optimistic for the CPU: the '
                  'cost of its LU grows like DoF^%.2f between the two largest '
                  'samples; measured at size offline: oracle_steps_at_size)'
                  % (big['workload'], big['dofs'], big['step_s'],
                     big['dofs_per_s'], ndofs, fit['exponent']),
        # NOT a measurement: the power law through the two largest samples,
        # evaluated at this workload's size
        'oracle_step_extrapolated': {
            'kind': 'extrapolation',
            'steps_per_s': 1.0 / fit['step_s_extrapolated'],
            'step_s': fit['step_s_extrapolated'],
            'note': 't ~ DoF^%.2f through %s; measured at size offline: '
                    'oracle_steps_at_size'
                    % (fit['exponent'], ' and '.join(fit['through']))},
        'oracle_step': steps,
        'oracle_step_fit': fit,
        # the same step with ITERATIVE solvers on the host cores (the Krylov
        # parts in C/OpenMP on all of them), measured on a 335 k-DoF channel
        'iterative_step': iterative,
        # the same oracle step MEASURED at 0.75 - 2.5 M DoF (offline, build
        # container: what the parity fixtures at size were computed with)
        'oracle_steps_at_size': {
            'kind': 'measured offline (build container, 1 core of 8): the '
                    'oracle steps behind tests/golden/ns_large_*.npz',
            'steps': oracle_steps_at_size()},
        # like for like on one sub-step: the pressure solve with the GPU's
        # algorithm on all host cores
        'pressure_solves_per_s': 1.0 / t_solve,
        'pressure_solve': {
            'cores': cores,
            'sample': '%d pressure solves (%.1f s) of the %d-row pressure-'
                      'Poisson system of this workload with the algorithm the '
                      'GPU runs (CG + smoothed-aggregation V(1,1) cycle, same '
                      'hierarchy, same right-hand side and stopping test; '
                      'oracle/cpu_cg.c, C/OpenMP, first-touched arrays): %d '
                      'iterations, %.1f ms per solve'
                      % (solves, solves * t_solve, n, its, 1e3 * t_solve),
            'converged': bool(ok),
            'mgcg_iterations': its,
            'mgcg_solve_ms': 1e3 * t_solve,
            'jacobi_cg_iterations_per_s': jac_rate,
            'spmv_GBps': spmv_bytes(n, A.nnz) / spmv_s / 1e9,
            'gpu_like_for_like': gpu,
            },
        }
    out['oracle_step_scaled'] = {
        'value': out['value'], 'unit': 'time-steps/s', 'cores': 1,
        'kind': 'port', 'sample': out['sample']}
    if 'dofs_per_s' in iterative:
        out['value'] = iterative['dofs_per_s'] / float(ndofs)
        out['cores'] = cores
        out['kind'] = 'restatement'
        out['sample'] = (
            'one Rotational step with iterative solvers on the host cores '
            '(oracle/cpu_step.py: numpy assembly -- the momentum forms chunked '
            'over %d threads --, Newton systems by SuperLU-ILU + GMRES(30) on '
            'ONE core, pressure by CG + the GPU path\'s V-cycle and the mass '
            'system by Jacobi-CG in C/OpenMP on %d cores) MEASURED on %s: '
            '%.1f s per step = %.0f DoF/s, scaled linearly with the DoF count '
            'to the %d DoF of this workload (optimistic for the CPU); phases '
            'in iterative_step.seconds' % (
                cores, cores, iterative['workload'], iterative['step_s'],
                iterative['dofs_per_s'], ndofs))
    return out, x_cpu


def cpu_iterative_step(nx=400, ny=93):
    '''One step on the HOST CORES with iterative solvers (oracle/cpu_step.py:
    the oracle's numpy assembly, Newton systems by ILU-preconditioned GMRES,
    pressure by CG + the GPU path's smoothed-aggregation V-cycle in C/OpenMP,
    mass system by Jacobi-CG in C/OpenMP), MEASURED on a body-fitted channel
    of nx x ny (a bounded sample of the workload: scipy's ILU and GMRES run on
    one core and take most of the time).  The hierarchy is the product's own
    (built on the GPU side for the sample mesh and handed over as data).'''
    import numpy
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import large_cases
    from oracle import cpu_lib, cpu_step
    from flow_amd import karman
    from flow_amd.fem.bcs import collect
    from flow_amd.fem.multigrid import Multigrid
    try:
        lib = cpu_lib.load(cpu_lib.build(native=True, out_dir='/tmp'))
    except Exception:                                  # noqa: BLE001
        lib = cpu_lib.load()
    cores = lib.oracle_num_threads()
    case = large_cases.KarmanStepCase(nx, ny)
    # the product's pressure operator and hierarchy for this mesh
    prob = karman.KarmanProblem(nx, ny)
    prob.prepare()
    lay = prob.P.layout
    Kbc = [v for k, v in lay._dev.items()
           if isinstance(k, tuple) and k[0] == 'K_bc'][0][0]
    dofs, _v = collect(prob.p_bcs, lay.N)
    isbc = numpy.zeros(lay.N, dtype=bool)
    isbc[dofs] = True
    mg = Multigrid(Kbc, isbc, keep_host=True)
    hier = cpu_lib.MgHierarchy(lib, mg.host_levels, mg.Ainv_host, mg.omega) \
        if mg.nlevels >= 2 else None
    W, P = case.oracle_spaces()
    u_bc, p_bc = case.bc_data()
    t0 = time.perf_counter()
    _u1, _p1, _ui, info = cpu_step.step(
        W, P, case.u0, case.p0, case.lattice(case.f0), case.lattice(case.f1),
        u_bc, p_bc, case.rho, case.mu, case.dt, lib, hierarchy=hier, tol=1e-10,
        assembly_threads=cores)
    wall = time.perf_counter() - t0
    return {
        'kind': 'restatement, measured',
        'workload': 'Karman channel %d x %d, P2-P1, %d DoF, one Rotational '
                    'step from the analytic state of tests/large_cases.py (%d '
                    'Newton iterations)' % (nx, ny, case.num_dofs(),
                                            len(info['newton_history']) - 1),
        'dofs': case.num_dofs(),
        'step_s': wall,
        'dofs_per_s': case.num_dofs() / wall,
        'cores': {'momentum assembly (numpy, chunked over threads)': cores,
                  'other assembly (numpy)': 1,
                  'ILU + GMRES (scipy / SuperLU)': 1,
                  'pressure (%s, C/OpenMP)' % info['pressure_solver']: cores,
                  'velocity correction (Jacobi-CG, C/OpenMP)': cores},
        'seconds': info['seconds'],
        'gmres_iterations': info['gmres_iterations'],
        'pressure_iterations': info['pressure_iterations'],
        'correction_iterations': info['correction_iterations'],
        }


def oracle_steps_at_size():
    '''Oracle steps MEASURED at size, offline: the fixtures of tests/golden/
    ns_large_*.npz (tests/golden/make_golden.py --large, build container, one
    core of 8) keep the wall time of every oracle step they were computed
    with.'''
    import glob
    import numpy
    rows = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden',
                                              'ns_large_*.npz'))):
        try:
            d = numpy.load(path)
            rows.append({
                'fixture': os.path.basename(path),
                'dofs': int(d['num_dofs']),
                'velocity_degree': int(d['arg_vdeg']),
                'step_s': {k[:-len('_oracle_seconds')]: float(d[k])
                           for k in d.files if k.endswith('_oracle_seconds')},
                'newton_iterations': len(d['backward_euler_newton_history']) - 1,
                })
        except Exception:                              # noqa: BLE001
            continue
    return rows


def oracle_step_timing(target_dofs=None, budget_s=25.0):
    '''Full `step()` of the CPU oracle (numpy/scipy, sparse LU for every solve:
    oracle/fem_oracle.py) at BASELINE config 1 (unit square, n = 8, P2-P1) and
    on body-fitted Karman channels of growing size, as far as `budget_s`
    seconds of CPU work allow; plus the power law through the two largest,
    evaluated at `target_dofs`.'''
    import math
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import cases
    from flow_amd import fem
    out = []
    spent = 0.0
    for name, make, kind in (
            ('C1 unit square 8x8 crossed',
             lambda: fem.UnitSquareMesh(8, 8, 'crossed'), 'all'),
            ('Karman channel 100x23',
             lambda: fem.karman_channel(100, 23, fitted=True), 'channel'),
            ('Karman channel 160x37',
             lambda: fem.karman_channel(160, 37, fitted=True), 'channel'),
            ('Karman channel 300x70',
             lambda: fem.karman_channel(300, 70, fitted=True), 'channel')):
        # (a bounded sample: the next size costs ~4x the one before)
        if out and spent + 4.0 * out[-1]['step_s'] > budget_s:
            break
        case = cases.Case(make(), vdeg=2, dt=0.01, bc_kind=kind, rho=1.0,
                          mu=0.05, seed=0)
        t0 = time.perf_counter()
        case.oracle_step('rotational')
        wall = time.perf_counter() - t0
        spent += wall
        ndof = case.W.size() + case.P.size()
        out.append({'workload': name, 'dofs': ndof, 'step_s': wall,
                    'dofs_per_s': ndof / wall})
    a, b = out[-2], out[-1]
    expo = math.log(b['step_s'] / a['step_s']) / math.log(
        float(b['dofs']) / a['dofs'])
    fit = {'exponent': expo, 'through': [a['workload'], b['workload']]}
    if target_dofs:
        fit['target_dofs'] = target_dofs
        fit['step_s_extrapolated'] = b['step_s'] * (
            float(target_dofs) / b['dofs'])**expo
    return out, fit


def launch_ranks(n):
    '''`python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>` as
    a child process; its stdout (rank 0's JSON line) is passed through.
    Returns the child's exit code.'''
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    last = None
    for raw in proc.stdout:
        line = raw.decode('utf-8', 'replace').rstrip('\n')
        if line.startswith('{') and '"metric"' in line:
            last = line
        else:
            # (anything else a rank wrote to stdout: not this program's line)
            sys.stderr.write(line + '\n')
    code = proc.wait()
    if last is not None:
        print(last)
        sys.stdout.flush()
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # The run starts at dt0 = 1e-5 and the controller at most doubles dt per
    # step.  That ramp is SETUP here (`KarmanProblem.settle`: steps until dt
    # has moved by < 1 % three times in a row, outside every timed region, the
    # settled state kept): whatever --warmup / --steps the caller passes, the
    # timed steps are CFL-sized steps with at least one Newton iteration each
    # -- the regime a Karman run spends its time in (asserted below).
    # `--no-settle --steps 3 --warmup 1` times the start-up transient instead.
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-settle', action='store_true',
                    help='time the start-up transient: windows start at the '
                         'initial state with dt = dt0 instead of at the '
                         'plateau of the CFL controller')
    ap.add_argument('--spin-up', type=int, default=0,
                    help='untimed steps between the settled plateau and the '
                         'windows (setup): 2400 of them reach the DEVELOPED '
                         'vortex street of the headline workload (t ~ 75), '
                         'where a step takes two Newton iterations; the '
                         'default window is the early plateau, as in rounds '
                         '2-3')
    ap.add_argument('--developed', type=int, default=None, metavar='STEPS',
                    help='N = 1: a second window in the DEVELOPED vortex '
                         'street -- that many untimed steps behind the settled '
                         'plateau (setup), then --warmup / --steps as for the '
                         'headline window; reported as config.developed and '
                         'value_developed.  Default: 2400 at the headline size '
                         '(t ~ 75: the wake sheds, a step takes two Newton '
                         'iterations), 0 = off otherwise')
    ap.add_argument('--developed-period', type=int, default=400, metavar='STEPS',
                    help='with --developed: that many further steps behind '
                         'the timed window, timed as a whole (about one '
                         'shedding period: bursts and calm phases averaged); '
                         'reported as value_developed_period; 0 = off')
    ap.add_argument('--headline', default='street', choices=['street', 'plateau'],
                    help="what `value` is: 'street' = the mean over one shedding "
                         'period of the developed vortex street (needs the '
                         "developed phase), 'plateau' = the K-step window on "
                         'the early plateau (rounds 1-5)')
    ap.add_argument('--nx', type=int, default=2182,
                    help='cells along the channel (2182 x 509: ~10 M DoF)')
    ap.add_argument('--ny', type=int, default=None)
    ap.add_argument('--scheme', default='rotational',
                    choices=['chorin', 'ipcs', 'rotational'])
    ap.add_argument('--tol', type=float, default=1.0e-10)
    ap.add_argument('--velocity-degree', type=int, default=2, choices=[1, 2],
                    help='2: P2-P1 Taylor-Hood (headline); 1: P1-P1 '
                         '(BASELINE config 1M DoF: --nx 1196 --velocity-degree 1)')
    ap.add_argument('--mode', default='parity', choices=['parity', 'fast'],
                    help="solver mode of the headline value (default 'parity': "
                         "the reference's Newton path)")
    ap.add_argument('--no-fast-leg', action='store_true',
                    help="N = 1: skip the second window in mode 'fast'")
    ap.add_argument('--newton', action='append', default=[],
                    metavar='KEY=VALUE',
                    help='override an entry of solver_parameters["newton"] '
                         '(development: e.g. linear_solver=bicgstab); '
                         'recorded in config')
    ap.add_argument('--mu', type=float, default=0.002,
                    help='dynamic viscosity (reference driver: 0.002); '
                         'development: a smaller mesh with mu scaled by the '
                         'mesh width keeps the cell Peclet number of the '
                         'headline workload')
    ap.add_argument('--dt0', type=float, default=1.0e-5,
                    help='initial step size (reference driver: 1e-5)')
    ap.add_argument('--initial', default=None,
                    choices=['profile', 'stokes'],
                    help="initial state: 'stokes' = flow_amd.stokes.solve as "
                         "the reference driver (tests/test_karman_vortex_street"
                         ".py:171-179; default for the Taylor-Hood pair); "
                         "'profile' = the inflow profile (default for P1-P1: "
                         "the pair is not inf-sup stable, the Stokes system "
                         "has no unique pressure to converge to)")
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend for N > 1: 'nccl' (= RCCL, "
                         "one GPU per rank); 'gloo' only to rehearse several "
                         "ranks on one GPU")
    ap.add_argument('--shard-single', action='store_true',
                    help='development: run the sharded loops on a 1-rank '
                         'process group (measures their host overhead)')
    ap.add_argument('--halos', default='allreduce', choices=['allreduce', 'peer'],
                    help="N > 1: 'allreduce' = halos travel in the one all-reduce "
                         "of the strips (default: the path every multi-rank "
                         "test runs); 'peer' = from neighbour to neighbour "
                         'through IPC-mapped landing buffers with sequence '
                         'flags (flow_peer; self-tested at start, every rank '
                         'falls back to the all-reduce if any rank fails it)')
    ap.add_argument('--weak', action='store_true',
                    help='weak scaling: the channel is N times as long (nx x N '
                         'columns of the same cells, the same ny), the work per '
                         'GPU stays that of the 1-GPU run; default: strong '
                         'scaling, the mesh is fixed')
    ap.add_argument('--stage-timeout', type=float, default=300.0,
                    help='seconds a bring-up stage (process group, first '
                         'collective, first time step) may take on a rank '
                         'before that rank gives up with exit code 3 and a '
                         'line that names rank and stage')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hbm-resident', action='store_true')
    ap.add_argument('--spmv-reps', type=int, default=100)
    args = ap.parse_args()

    if args.initial is None:
        args.initial = 'stokes' if args.velocity_degree == 2 else 'profile'
    if args.developed is None:
        args.developed = 2400 if (
            args.nx == 2182 and args.ny is None and args.velocity_degree == 2
            and not args.no_settle and not args.spin_up
            and args.mode == 'parity' and args.headline == 'street') else 0
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1 and args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` as written: start the N ranks as a CHILD
        # process (never exec: nothing in this process has touched torch or
        # the GPU yet, and nothing will), relay its one JSON line and its exit
        # code.  The driver's own launch line (python -m torch.distributed.run
        # ... bench.py --gpus N) sets RANK / WORLD_SIZE and comes in below.
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit('--gpus %d, but WORLD_SIZE = %d' % (args.gpus, world))
    import numpy
    import torch
    import torch.distributed as dist
    from flow_amd import device, _hip, karman, parallel
    from flow_amd.fem import ops
    from flow_amd.fem.bcs import collect
    import flow_amd.navier_stokes as navsto

    _hip.lib()          # fail loudly without the HIP library / a GPU

    # -- bring-up that cannot hang a lease: every rank says which stage it is
    # in (stderr), and a stage of the bring-up that takes longer than
    # --stage-timeout ends THAT rank with exit code 3 and a line naming rank
    # and stage (os._exit from a watchdog thread: never an exec, and a hung
    # collective cannot be interrupted any other way); the launcher then ends
    # the others.
    import threading
    stage = {'name': 'start', 'timer': None}

    def enter(name, guarded=False, limit=None):
        if stage['timer'] is not None:
            stage['timer'].cancel()
            stage['timer'] = None
        stage['name'] = name
        if world > 1 or args.shard_single:
            sys.stderr.write('[bench rank %d/%d] stage: %s\n' % (rank, world, name))
            sys.stderr.flush()
        if guarded and world > 1:
            seconds = args.stage_timeout if limit is None else limit

            def give_up():
                sys.stderr.write(
                    '[bench rank %d/%d] FAILED: stage %r did not finish in '
                    '%.0f s -- giving up (exit code 3)\n'
                    % (rank, world, name, seconds))
                sys.stderr.flush()
                os._exit(3)
            t = threading.Timer(seconds, give_up)
            t.daemon = True
            t.start()
            stage['timer'] = t

    class stdout_to_stderr(object):
        '''RCCL prints a version banner on STDOUT when a communicator comes
        up; this program's stdout is ONE JSON line.'''
        def __enter__(self):
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)

        def __exit__(self, *exc):
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)

    collective_us = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        with stdout_to_stderr():
            enter('process group (%s)' % args.backend, guarded=True)
            if args.backend == 'nccl':
                dist.init_process_group('nccl', device_id=device.get())
            else:
                dist.init_process_group('gloo')
            enter('first collective', guarded=True)
            probe = torch.ones(1, dtype=torch.float64,
                               device=device.get() if args.backend == 'nccl'
                               else 'cpu')
            dist.all_reduce(probe)
            if args.backend == 'nccl':
                torch.cuda.synchronize()
            if int(round(float(probe.item()))) != world:
                sys.stderr.write('[bench rank %d/%d] FAILED: first collective '
                                 'summed to %r\n' % (rank, world, probe.item()))
                sys.exit(4)
            enter('communicator of the strips', guarded=True)
            parallel.enable(dist.group.WORLD)
            # torch.distributed's all-reduce or the one the library issues
            # itself (csrc/rccl_direct.hip): 50 calls of each, the faster one
            # is used -- every rank takes the same decision
            enter('collective micro-benchmark', guarded=True)
            collective_us = parallel.comm().use_fastest(50)
            if args.halos == 'peer':
                enter('peer halos: map + self-test', guarded=True)
                collective_us['halos'] = 'peer' \
                    if parallel.comm().enable_peer() else \
                    'allreduce (the peer self-test failed)'
            else:
                collective_us['halos'] = 'allreduce'
            enter('setup')
    elif args.shard_single:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        with stdout_to_stderr():
            dist.init_process_group(args.backend, rank=0, world_size=1,
                                    **({'device_id': device.get()}
                                       if args.backend == 'nccl' else {}))
            parallel.enable(dist.group.WORLD, force=True)
            # (the micro-benchmark of the N > 1 bring-up, on the 1-rank group:
            # exercises both bindings where RCCL can run on this box)
            collective_us = parallel.comm().use_fastest(50)
            collective_us['halos'] = 'allreduce'

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_setup = time.perf_counter()
    ny = args.ny if args.ny else max(2, int(round(args.nx * 509.0 / 2182.0)))
    nx_run, length = args.nx, karman.X1
    if args.weak and world > 1:
        nx_run, length = args.nx * world, karman.X1 * world
    prob = karman.KarmanProblem(nx_run, ny,
                                velocity_degree=args.velocity_degree,
                                scheme=args.scheme, mu=args.mu, length=length)

    start = {}
    settled = {}

    def initial_state():
        prob.reset(args.dt0)
        if args.initial == 'stokes':
            # the reference driver's start (tests/test_karman_vortex_street.py
            # :171-179); solved once, outside every timed region (replicated on
            # every rank: setup)
            if not start:
                prob.set_initial_stokes()
                start['u'] = _hip.clone(prob.u0.data)
                start['p'] = _hip.clone(prob.p0.data)
                start['info'] = dict(prob.stokes_info)
            ops.copy(prob.u0.data, start['u'])
            ops.copy(prob.p0.data, start['p'])
        else:
            prob.set_initial_profile()

    def apply_overrides():
        # --newton KEY=VALUE; KEY may be dotted (pmg.coarse_steps=3) and may
        # start with another group of solver_parameters (correction.method=cg)
        for kv in args.newton:
            key, val = kv.split('=', 1)
            path = key.split('.')
            d = navsto.solver_parameters
            if path[0] not in d:
                d = d['newton']
            for part in path[:-1]:
                d = d[part]
            old = d[path[-1]]
            d[path[-1]] = val if isinstance(old, str) else type(old)(float(val))

    def window(mode):
        '''W warm-up + K timed steps from the initial state in `mode`.'''
        navsto.set_mode(mode)
        apply_overrides()
        if settled:
            prob.restore(settled['state'])
        else:
            initial_state()
        for _ in range(args.warmup):
            prob.step(tol=args.tol)
        # (marker launches 1 / 2 bracket the timed steps in a kernel trace:
        # profiles/summarize.py cuts there; outside the clock)
        _hip.check(_hip.lib().flow_profile_marker(1, _hip.stream()))
        barrier()
        calls0 = parallel.comm().calls if parallel.active() else 0
        launches0 = _hip.launch_count()
        graphs0 = _hip.graph_stats()
        t0 = time.perf_counter()
        infos = [prob.step(tol=args.tol) for _ in range(args.steps)]
        barrier()
        elapsed = time.perf_counter() - t0
        infos[0]['launches_in_window'] = _hip.launch_count() - launches0
        graphs1 = _hip.graph_stats()
        infos[0]['graphs_in_window'] = {
            k: graphs1[k] - graphs0[k] for k in (
                'captures', 'replays', 'nodes', 'captures_cg', 'captures_gmres',
                'captures_mass')}
        if parallel.active():
            # (every halo and every reduction of the strips is one all-reduce)
            infos[0]['collectives_in_window'] = parallel.comm().calls - calls0
        _hip.check(_hip.lib().flow_profile_marker(2, _hip.stream()))
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64,
                              device=device.get())
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return infos, elapsed

    def summary(infos, elapsed):
        tim = {}
        for key in ('tentative_s', 'pressure_s', 'correction_s'):
            tim[key] = sum(i.get('timings', {}).get(key, 0.0) for i in infos) \
                / len(infos)
        return {
            'steps_per_s': len(infos) / elapsed,
            'ms_per_step': 1.0e3 * elapsed / len(infos),
            'dt': [i['dt'] for i in infos],
            'newton_residuals': [i['newton_residuals'] for i in infos],
            'newton_iterations': [len(i['newton_residuals']) - 1
                                  for i in infos],
            'newton_linear_applications': [
                sum(i.get('newton_linear_applications', [])) for i in infos],
            'newton_start': [i.get('initial_guess', 'u0') for i in infos],
            'pressure_cg_iterations': [i['pressure'].iterations for i in infos],
            'correction_cg_iterations': [i['correction'].iterations
                                         for i in infos],
            'cfl_projection_cg_iterations': [
                i.get('projection_iterations', 0) for i in infos],
            'substep_s': tim,
            # kernel launches of the library per step (flow_launch_count)
            'launches_per_step': infos[0].get('launches_in_window', 0)
            / float(len(infos)),
            # iteration bodies replayed as HIP graphs in the window (off by
            # default: FLOW_AMD_GRAPHS=1; a replay counts as ONE launch above)
            'graph_replay': infos[0].get('graphs_in_window'),
            # (counted at the all-reduce callback; the library-issued
            # ncclAllReduce of FLOW_AMD_RCCL_DIRECT=1 does not pass there)
            'collectives_per_step': (
                infos[0]['collectives_in_window'] / float(len(infos))
                if infos[0].get('collectives_in_window') else None),
            }

    # everything that is built once and cached (operators, hierarchy, ILU
    # plan, ...), then the start state: all of it setup, outside the windows
    # (development overrides first: some of them shape those structures)
    apply_overrides()
    enter('first time step (prepare)', guarded=True)
    prob.prepare()
    enter('initial state')
    initial_state()
    if not args.no_settle:
        navsto.set_mode(args.mode)
        settled['steps'] = prob.settle(tol=args.tol)
        for _ in range(args.spin_up):
            prob.step(tol=args.tol)
        settled['spin_up'] = args.spin_up
        settled['state'] = prob.snapshot()
    barrier()
    setup_s = time.perf_counter() - t_setup
    enter('plateau window')
    infos, elapsed = window(args.mode)
    head = summary(infos, elapsed)
    if settled:
        # the contract of the settle phase: plateau steps only
        dts = head['dt']
        assert min(head['newton_iterations']) >= 1, head['newton_iterations']
        # (no step of the controller's start-up ramp, which doubles dt: the step
        # size moves by a few per cent per step at most)
        assert args.spin_up or \
            all(0.95 < b / a < 1.05 for a, b in zip(dts, dts[1:])), dts

    # --- pressure-Poisson SpMV against the HBM roofline (dominant kernel) ---
    P = prob.P
    lay = P.layout
    Kbc = [v for k, v in lay._dev.items()
           if isinstance(k, tuple) and k[0] == 'K_bc'][0][0]
    n, nnz = lay.N, lay.nnz
    bytes_alg = spmv_bytes(n, nnz)
    # (a) as the CG runs it (on the strips: this rank's rows of the matrix),
    # (b) back-to-back replay of the whole-matrix launch
    t_solver, launches, t_raw, t_over = measure_spmv_in_solver(
        prob, n, 3, args.tol)
    t_replay = measure_spmv_replay(Kbc.apply, n, reps=args.spmv_reps)
    bytes_solver = bytes_alg
    if parallel.active():
        pv = parallel.view(lay)
        rp = lay.pattern('rowptr')
        bytes_solver = spmv_bytes(pv.r1 - pv.r0,
                                  int(rp[pv.r1]) - int(rp[pv.r0]))
    if not launches:
        t_solver, bytes_solver = t_replay, bytes_alg
    achieved = bytes_solver / t_solver / 1e9
    traffic = None
    traffic_source = None
    # (PMC summary of the same in-solver dispatches, written by
    # profiles/run_profiles.sh from separate --pmc passes of this command: a
    # CITATION of the newest committed summary, not something this run measures)
    import glob
    tfiles = sorted(glob.glob(os.path.join(
        ROOT, 'profiles', 'spmv_traffic_r[0-9][0-9]_in_solver.json')))
    # the committed PMC summary belongs to the headline workload only
    if tfiles and args.nx == 2182 and args.ny is None:
        try:
            traffic = json.load(open(tfiles[-1])).get('hbm_bytes_per_launch')
            traffic_source = (
                'profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                'this bench command (profiles/run_profiles.sh), median over the '
                'in-solver dispatches; committed with the round it names, not '
                'measured by this run' % os.path.basename(tfiles[-1]))
        except Exception:                              # noqa: BLE001
            traffic = None
    copy_ceiling, read_ceiling = measure_stream_ceilings() if world == 1 \
        else (None, None)
    resident = None
    if world == 1 and not args.no_hbm_resident:
        resident = measure_spmv_hbm_resident()

    fast = None
    zero_start = None
    if world == 1 and args.mode == 'parity' and not args.no_fast_leg:
        f_infos, f_elapsed = window('fast')
        fast = summary(f_infos, f_elapsed)
        navsto.set_mode('parity')
        # the same window with every Krylov solve started as a single call
        # would start it (nothing but preconditioners carried from step to
        # step): what the start vectors extrapolated in time are worth
        saved = {g: dict(navsto.solver_parameters[g])
                 for g in ('newton', 'pressure', 'correction')}
        args.newton = list(args.newton) + [
            'linear_start=zero', 'pressure.start=zero',
            'correction.increment_start=zero']
        z_infos, z_elapsed = window('parity')
        zero_start = summary(z_infos, z_elapsed)
        args.newton = args.newton[:-3]
        for g, vals in saved.items():
            navsto.solver_parameters[g].update(vals)

    # --- the developed vortex street (N = 1): the regime the metric names ---
    # The windows above sit on the early plateau, a few steps behind the CFL
    # controller's ramp: symmetric flow, one Newton iteration per step.  Once
    # the wake sheds (t > ~55) ||F(u0)|| is 1e-8 instead of 5e-10, the first
    # Newton iterate is left just above `tol`, every step takes two Newton
    # iterations and longer solves -- for the reference as for this build.
    developed = None
    developed_error = None
    if args.developed > 0 and settled:
        # (N > 1: the street has only ever run on strips in short gloo
        # rehearsals -- a solver failure there, which every rank sees alike,
        # must not cost the run its line: `value` then falls back to the
        # plateau window and says so; a rank that hangs is ended by the
        # watchdog, 6 x the bring-up limit)
        enter('developed street: %d spin-up steps' % args.developed,
              guarded=True, limit=6.0 * args.stage_timeout)
    try:
        if not (args.developed > 0 and settled):
            raise StopIteration
        navsto.set_mode(args.mode)
        apply_overrides()
        prob.restore(settled['state'])
        t_spin = time.perf_counter()
        for _ in range(args.developed):
            prob.step(tol=args.tol)
        device.synchronize()
        spin_s = time.perf_counter() - t_spin
        plateau_state = settled['state']
        settled['state'] = prob.snapshot()
        # (restore() forgets the start-vector histories: the window's warm-up
        # steps refill them; at least 8, so that the timed steps see full ones)
        keep_warmup = args.warmup
        args.warmup = max(args.warmup, 8)
        d_infos, d_elapsed = window(args.mode)
        args.warmup = keep_warmup
        developed = summary(d_infos, d_elapsed)
        developed['spin_up_steps'] = args.developed
        developed['spin_up_s'] = spin_s
        developed['t'] = settled['state']['t']
        developed['unorm'] = [i.get('unorm') for i in d_infos]
        developed['note'] = (
            '%d untimed steps behind the settled plateau (setup, %.1f s), then '
            '%d warm-up + %d timed steps in mode %r: the developed vortex '
            'street' % (args.developed, spin_s, max(args.warmup, 8),
                        args.steps, args.mode))
        # ... and one shedding period on from there (St ~ 0.2: ~12.5 s, ~370
        # steps): what a step of the street costs averaged over its bursts and
        # its calm phases -- the window above happens to sit in a burst
        if args.developed_period > 0:
            barrier()
            t_p = time.perf_counter()
            p_infos = [prob.step(tol=args.tol)
                       for _ in range(args.developed_period)]
            barrier()
            p_elapsed = time.perf_counter() - t_p
            if world > 1:
                tt = torch.tensor([p_elapsed], dtype=torch.float64,
                                  device=device.get())
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                p_elapsed = float(tt.item())
            apps = [sum(i.get('newton_linear_applications', [])) for i in p_infos]
            developed['period'] = {
                'steps': args.developed_period,
                'ms_per_step': 1e3 * p_elapsed / args.developed_period,
                'steps_per_s': args.developed_period / p_elapsed,
                't_end': prob.t,
                'newton_iterations_mean': sum(
                    len(i['newton_residuals']) - 1 for i in p_infos)
                / float(len(p_infos)),
                'newton_linear_applications_mean': sum(apps) / float(len(apps)),
                'newton_linear_applications_min_max': [min(apps), max(apps)],
                'pressure_cg_iterations_mean': sum(
                    i['pressure'].iterations for i in p_infos)
                / float(len(p_infos)),
                'note': 'the %d steps behind the timed window, continued '
                        'without a restart: one shedding period'
                        % args.developed_period,
                }
        settled['state'] = plateau_state
    except StopIteration:
        pass
    except RuntimeError as exc:
        if world == 1:
            raise
        # (a solver verdict: the same on every rank of the strips)
        developed = None
        developed_error = repr(exc)
        sys.stderr.write('[bench rank %d/%d] the developed street failed: %s -- '
                         '`value` falls back to the plateau window\n'
                         % (rank, world, developed_error))

    enter('report')
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # what `value` is: the street's period mean where the developed phase ran
    street = developed is not None and 'period' in developed \
        and args.headline == 'street'
    headline = developed['period'] if street else head
    out = {
        'metric': 'ipcs_time_steps_per_sec',
        'value': headline['steps_per_s'],
        'unit': 'time-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': headline['ms_per_step'],
        # the steps `value` was timed over (one shedding period, timed as a
        # whole between two barriers) -- `steps` is the K of the command line,
        # the length of the windows value_developed / value_plateau
        'headline_steps': developed['period']['steps'] if street else args.steps,
        'value_plateau': head['steps_per_s'],
        'ms_per_step_plateau': head['ms_per_step'],
        'higher_is_better': True,
        'scaling': 'weak' if (args.weak and world > 1) else 'strong',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': 'Karman vortex street %s, %d DoF '
                        '(%d x %d structured channel, body-fitted cylinder), '
                        '%s scheme, backward Euler, tol %.0e, mu %g, '
                        'rho 998.2, dt0 1e-5 + CFL controller, start: %s; '
                        'value: %s'
                        % ('P2-P1 Taylor-Hood' if args.velocity_degree == 2
                           else 'P1-P1', prob.num_dofs(), nx_run, ny,
                           args.scheme, args.tol, args.mu,
                           'Stokes solution' if args.initial == 'stokes'
                           else 'inflow profile',
                           'the developed vortex street, mean over one shedding '
                           'period (config.headline)' if street
                           else 'the early plateau of the run (config.headline)'),
            'headline': (
                'value = mean over one shedding period of the DEVELOPED VORTEX '
                'STREET: %d untimed steps behind the settled plateau, %d + %d '
                'steps (value_developed: that K-step window, in a burst of the '
                'run), then %d steps timed as a whole, t = %.1f .. %.1f s; '
                'value_plateau = the K-step window on the early symmetric '
                'plateau (the `value` of rounds 1-5)' % (
                    developed['spin_up_steps'], max(args.warmup, 8), args.steps,
                    developed['period']['steps'], developed['t'],
                    developed['period']['t_end'])) if street else (
                'value = the K-step window on the early plateau (symmetric '
                'flow, one Newton iteration per step)%s' % (
                    '' if args.headline == 'plateau' else
                    ': the developed phase FAILED on the strips (%s)'
                    % developed_error if developed_error else
                    ': the developed phase did not run (--developed 0 / not the '
                    'headline size)')),
            'mode': args.mode,
            'mode_note': "headline = mode '%s'%s" % (
                args.mode,
                " (Newton from u0, linear residual <= 1e-6 of the Newton "
                "tolerance: the reference's path; one step matches an exact "
                "Newton step to < 1e-6 in u and p, "
                "tests/test_full_size_parity.py)" if args.mode == 'parity'
                else ''),
            'dofs': prob.num_dofs(),
            'cells': prob.mesh.num_cells(),
            'pressure_rows': n,
            'pressure_nnz': nnz,
            # (dispatch size of the pressure SpMV: profiles/run_profiles.sh
            # picks its launches out of the PMC passes by it)
            'pressure_spmv_grid': int(Kbc.operator().nblocks) * 256,
            'parallelism': parallel.describe(world, n),
            # microseconds per all-reduce of 8 doubles (50 back-to-back calls,
            # slowest rank) through torch.distributed and through the
            # library's own ncclAllReduce, and which of them the run used
            'collective_us': collective_us,
            'setup_s': setup_s,
            'settle': {
                'steps': settled['steps'], 'dt': settled['state']['dt'],
                't': settled['state']['t'], 'spin_up': settled['spin_up'],
                'note': 'steps taken before the windows (setup) until the CFL '
                        "controller's dt moved by < 1 % three times in a row; "
                        'every window starts from that state'}
            if settled else None,
            'newton_iterations_min': min(head['newton_iterations']),
            'stokes_start': start.get('info'),
            # (what the last timed step actually ran with: the p-multigrid
            # cycle, or the ILU(0) where that is rejected / not applicable)
            'newton_linear_solver': navsto.solver_parameters['newton'].get(
                'linear_solver', 'gmres') + '+' + str(
                    infos[-1].get('newton_preconditioner', 'none')),
            'pmg_contraction': next(
                (pre.contraction for name, pre in prob.W.layout._dev.items()
                 if name in ('jacobian_pmg', 'jacobian_pmg_strip')
                 and hasattr(pre, 'contraction')), None),
            'newton_overrides': args.newton,
            # every Krylov vector, matrix, dot product and update is fp64; the
            # PRECONDITIONERS of the flexible GMRES keep their own copies of
            # the matrix in reduced precision (p-multigrid: row-scaled fp16
            # entries, fp32 vectors; ILU(0): fp32 factors and sweep vector,
            # fp64 row sums) -- they only steer the Krylov path, the converged
            # step is the same (tests/test_full_size_parity.py; DESIGN.md
            # section 3; profiles/NOTES.md section 4)
            'preconditioner_storage': {
                'pmg_matrix': 'fp16 (row-scaled)', 'pmg_vectors': 'fp32',
                'ilu_factors': navsto.solver_parameters['newton'].get(
                    'ilu_storage', 'fp64'),
                'ilu_sweep_vector': navsto.solver_parameters['newton'].get(
                    'ilu_vector', 'fp64')},
            },
        'roofline': {
            'kernel': 'spmv_stream_kernel<DOT> (pressure-Poisson CSR SpMV '
                      'with the fused z.Az partials, fp64), as the pressure '
                      'CG launches it inside the time steps'
                      if launches else
                      'spmv_stream_kernel (pressure-Poisson CSR SpMV, fp64), '
                      'back-to-back replay',
            'bound': 'hbm',
            'achieved': achieved,
            'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBPS,
            'traffic': traffic if not parallel.active() else None,
            'traffic_source': traffic_source if not parallel.active() else None,
            # the box's own ceilings, timed in this run: a plain
            # streaming copy of 1 GiB in + 1 GiB out, and a kernel that only
            # reads 1 GiB (the SpMV is read-dominated: 8 % of its bytes are
            # stores)
            'copy_ceiling_GBps': copy_ceiling,
            'read_ceiling_GBps': read_ceiling,
            'frac_of_copy': achieved / copy_ceiling if copy_ceiling else None,
            'frac_of_read': achieved / read_ceiling if read_ceiling else None,
            'bytes_per_launch': bytes_solver,
            'us_per_launch': t_solver * 1e6,
            'launches_timed': launches,
            'timing': 'HIP events around every in-solver launch on the launch '
                      'stream (%.2f us per pair; the same pair around a null '
                      'kernel reads %.2f us of dispatch latency, not '
                      'subtracted)' % (t_raw * 1e6, t_over * 1e6),
            'warm_replay': {
                'us_per_launch': t_replay * 1e6,
                'GBps': bytes_alg / t_replay / 1e9,
                'frac': bytes_alg / t_replay / 1e9 / HBM_PEAK_GBPS,
                'note': 'the same matrix, %d back-to-back launches: the '
                        '114 MB working set then stays in the 256 MB '
                        'Infinity Cache' % args.spmv_reps,
                },
            'hbm_resident': resident,
            },
        }
    out['config'].update({k: v for k, v in head.items()
                          if k not in ('steps_per_s', 'ms_per_step')})
    if developed is not None:
        # co-headline: the same metric in the regime a Karman run lives in
        out['value_developed'] = developed['steps_per_s']
        out['ms_per_step_developed'] = developed['ms_per_step']
        if 'period' in developed:
            out['value_developed_period'] = developed['period']['steps_per_s']
            out['ms_per_step_developed_period'] = developed['period']['ms_per_step']
        out['config']['developed'] = developed
    if fast is not None:
        out['config']['fast_mode'] = fast
    if zero_start is not None:
        out['config']['zero_start'] = zero_start
    out['config']['start_vectors'] = (
        "mode 'parity' starts the Newton linear solve, the pressure CG and the "
        "velocity correction from the previous steps' increments extrapolated "
        "in time (least-squares cubic through five; navier_stokes.solver_parameters: "
        "linear_start / start / increment_start): start vectors only -- every "
        "solve converges to the same stopping test, the Newton iteration still "
        "starts from u0; 40-step trajectories agree with the zero-start run to "
        "4e-10 (u) / 1.3e-9 (p), tools/linear_start_check.py; the zero-start "
        "window is reported in config.zero_start")
    if world == 1 and not args.no_cpu_baseline:
        # the right-hand side of a real pressure solve is not kept; a fixed
        # synthetic one of the same smoothness class stands in on both sides
        dofs, _vals = collect(prob.p_bcs, n)
        isbc = numpy.zeros(n, dtype=bool)
        isbc[dofs] = True
        xs = lay.dof_coords
        b = numpy.sin(40.0 * xs[:, 0]) * numpy.cos(60.0 * xs[:, 1])
        b[isbc] = 0.0
        bd = device.to_device(b)
        mg = [v for k, v in lay._dev.items()
              if isinstance(k, tuple) and k[0] == 'mg'][0]
        dinv = Kbc.diag_inv()
        gpu = {}
        for name, kw in (('mgcg', {'mg': mg}), ('jacobi_cg', {})):
            xd = device.zeros(n)
            device.synchronize()
            t0 = time.perf_counter()
            sol = ops.krylov_solve('cg', Kbc, bd, xd, rtol=args.tol,
                                   maxit=200000, dinv=dinv,
                                   check_every=2 if kw else 200, **kw)
            device.synchronize()
            wall = time.perf_counter() - t0
            gpu[name] = {'iterations': sol.iterations, 'solve_ms': 1e3 * wall,
                         'iterations_per_s': sol.iterations / wall}
            if name == 'mgcg':
                x_gpu = device.to_host(xd).numpy()
        gpu['spmv_GBps_in_solver'] = achieved
        out['cpu_baseline'], x_cpu = cpu_baseline(Kbc, isbc, b, args.tol, gpu,
                                                  prob.num_dofs())
        out['cpu_baseline']['solution_rel_l2_gpu_vs_cpu'] = float(
            numpy.linalg.norm(x_gpu - x_cpu) / numpy.linalg.norm(x_cpu))
    print(json.dumps(out))
    sys.stdout.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
