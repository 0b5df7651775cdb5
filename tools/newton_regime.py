import os, sys
sys.path.insert(0, '/root/repo')
from flow_amd import karman
import flow_amd.navier_stokes as navsto
for af in ('previous', 'best'):
    navsto.solver_parameters['newton']['initial_guess'] = af
    prob = karman.KarmanProblem(2182, 509, velocity_degree=2)
    prob.set_initial_profile(); prob.dt = 1e-5
    import time
    for k in range(120):
        if k == 100:
            import torch; torch.cuda.synchronize(); t0 = time.time()
        info = prob.step(tol=1e-10)
    torch.cuda.synchronize()
    print('initial_guess', af, 'ms/step %.1f' % ((time.time() - t0) / 20 * 1e3), info.get('initial_guess'), info['newton_linear_iterations'], ['%.2e' % r for r in info['newton_residuals']], flush=True)
