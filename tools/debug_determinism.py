# Development: is one Karman step bitwise reproducible?  Runs the same step
# several times from the same state and compares the fields.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import karman
import flow_amd.navier_stokes as navsto

SIZE = (96, 24, 2)
args = []
for a in sys.argv[1:]:
    if a.startswith('size='):
        SIZE = tuple(int(v) for v in a[5:].split(','))
    else:
        args.append(a)
for key, val in [a.split('=') for a in args]:
    sec, name = key.split('.')
    navsto.solver_parameters[sec][name] = eval(val)
print(navsto.solver_parameters['newton'])
res = []
for trial in range(4):
    prob = karman.KarmanProblem(SIZE[0], SIZE[1], velocity_degree=SIZE[2])
    prob.set_initial_profile()
    prob.dt = 1e-5
    infos = [prob.step(tol=1e-10) for _ in range(3)]
    ui = navsto.last_step_info.get('tentative_velocity')
    res.append((prob.u0.array().copy(), prob.p0.array().copy(),
                [(i['newton_linear_iterations'], i['pressure'].iterations, i['correction'].iterations) for i in infos]))
for k in range(1, len(res)):
    du = abs(res[k][0] - res[0][0]).max()
    dp = abs(res[k][1] - res[0][1]).max()
    print('trial %d vs 0: max|du| %.3e  max|dp| %.3e  its %r vs %r' % (
        k, du, dp, res[k][2], res[0][2]))
if os.environ.get('DUMP'):
    numpy.savez(os.environ['DUMP'], u=res[0][0], p=res[0][1])
