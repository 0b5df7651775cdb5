# Development: repeat ONE step from the same state with the same settings and
# print how ui, p1, u1 move from run to run (they must not).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
from flow_amd import karman, device
import flow_amd.navier_stokes as navsto
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2182
prob = karman.KarmanProblem(nx, int(round(nx * 509.0 / 2182.0)))
prob.set_initial_profile()
npar = navsto.solver_parameters['newton']
for _ in range(2):
    prob.step()
u_s = prob.u0.vector().get_local().copy()
p_s = prob.p0.vector().get_local().copy()
dt, t = prob.dt, prob.t
npar.update(initial_guess='previous', forcing=0.0, adaptive_forcing=False)
navsto.solver_parameters['pressure']['extrapolate'] = False
navsto.solver_parameters['correction']['extrapolate'] = False
rel = lambda a, b: numpy.linalg.norm(a - b) / numpy.linalg.norm(b)
ref = None
for k, factor in enumerate([1e-9, 1e-9, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-9]):
    npar['linear_atol_factor'] = factor
    prob.u0.vector().set_local(u_s)
    prob.p0.vector().set_local(p_s)
    prob.dt, prob.t = dt, t
    info = prob.step(adapt=False)
    ui = navsto.last_step_info['tentative_velocity'].vector().get_local().copy()
    u1 = prob.u0.vector().get_local().copy()
    p1 = prob.p0.vector().get_local().copy()
    pre = prob.W.layout._dev.get('jacobian_ilu')
    if ref is None:
        ref = (ui, p1, u1)
    print('run %d factor %.0e: dui %.2e dp1 %.2e du1 %.2e  apps %r res %r lin_res %.2e ilu dt %.2e stale %s corr %r'
          % (k, factor, rel(ui, ref[0]), rel(p1, ref[1]), rel(u1, ref[2]),
             info['newton_linear_applications'], ['%.2e' % r for r in info['newton_residuals']],
             0.0, pre.dt, pre.stale, info['correction']), flush=True)
