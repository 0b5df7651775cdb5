# -*- coding: utf-8 -*-
'''Sharded pressure solve of the 2.5 M-DoF start-up steps (the scenario of
tests/test_parallel_gpu.py::test_strips_on_a_2M_dof_channel) with the two- and
the three-collective form of the V-cycle CG: iteration counts and residuals.
  FLOW_AMD_MGCG_COLLECTIVES=2|3 python tools/strip_pressure_check.py'''
from __future__ import print_function
import os
import socket
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp                      # noqa: E402


def worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOCAL_RANK'] = '0'
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from flow_amd import parallel, karman
        parallel.enable(dist.group.WORLD, force=True)
        prob = karman.KarmanProblem(1091, 254)
        prob.set_initial_profile()
        rows = []
        for k in range(int(os.environ.get('STEPS', '4'))):
            try:
                info = prob.step()
                rows.append((info['dt'], info['pressure'].iterations,
                             info['pressure'].residual,
                             sum(info['newton_linear_applications']),
                             info['correction'].iterations))
            except RuntimeError as e:
                rows.append(('FAILED', str(e)[-120:]))
                break
        if rank == 0:
            out[0] = rows
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    manager = mp.get_context('spawn').Manager()
    out = manager.dict()
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    print('collectives per V-cycle CG iteration: %s'
          % os.environ.get('FLOW_AMD_MGCG_COLLECTIVES', '2'))
    for r in out[0]:
        print('  ', r)
