# -*- coding: utf-8 -*-
'''GMRES applications per step with / without the extrapolated start vector of
the Newton linear solve, single GPU and 2 gloo ranks sharing the GPU.
  python tools/strip_linear_start.py [nx] [steps]'''
from __future__ import print_function
import os
import socket
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp                      # noqa: E402


def steps(nx, nsteps, mode):
    from flow_amd import karman
    import flow_amd.navier_stokes as navsto
    navsto.solver_parameters['pressure']['mg_coarsest'] = 200
    navsto.solver_parameters['newton']['linear_start'] = mode
    prob = karman.KarmanProblem(nx, max(2, nx // 4), mu=0.036)
    prob.set_initial_profile()
    prob.dt = prob.hmax / 0.016
    infos = [prob.step(adapt=False) for _ in range(nsteps)]
    return [sum(i['newton_linear_applications']) for i in infos], \
        [i['newton_preconditioner'] for i in infos][-1]


def worker(rank, world, port, nx, nsteps, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOCAL_RANK'] = '0'
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from flow_amd import parallel
        parallel.enable(dist.group.WORLD, force=True)
        res = {m: steps(nx, nsteps, m) for m in ('zero', 'extrapolated')}
        if rank == 0:
            out[0] = res
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for m in ('zero', 'extrapolated'):
        print('single GPU, %-12s: %r' % ((m,) + (steps(nx, nsteps, m),)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(worker, args=(2, port, nx, nsteps, out), nprocs=2, join=True)
    for m, v in out[0].items():
        print('2 strips,   %-12s: %r' % (m, v))
