# Development: BASELINE configs 4 and 5 at their nominal sizes, timed.
#   python tools/run_configs.py stokes|boussinesq [n]
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import fem, stokes, boussinesq, device

what = sys.argv[1]
if what == 'stokes':
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 470
    mesh = fem.UnitSquareMesh(n, n)
    WP = fem.FunctionSpace(
        mesh, fem.VectorElement('Lagrange', mesh.ufl_cell(), 2)
        * fem.FiniteElement('Lagrange', mesh.ufl_cell(), 1))
    W, P = WP.sub(0), WP.sub(1)

    class Lid(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[1] > 1.0 - 1e-12)

    class Walls(fem.SubDomain):
        def inside(self, x, on_boundary):
            return on_boundary & (x[1] <= 1.0 - 1e-12)

    class Corner(fem.SubDomain):
        def inside(self, x, on_boundary):
            return (x[0] < 1e-12) & (x[1] < 1e-12)
    bcs = [fem.DirichletBC(W, (0.0, 0.0), Walls()),
           fem.DirichletBC(W, (1.0, 0.0), Lid())]
    print('dofs', W.size() + P.N, flush=True)
    t = time.time()
    u, p = stokes.solve(WP, bcs, 1.0, fem.Constant((0.0, 0.0)), verbose=False,
                        tol=1e-8, max_iter=2000)
    device.synchronize()
    print('stokes solve %.1f s' % (time.time() - t), stokes.last_solve_info)
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    t = time.time()
    u1, p1, th, steps = boussinesq.compute_boussinesq(target_time=0.05, nx=n)
    device.synchronize()
    print('boussinesq %d steps %.1f s, dofs %d' % (
        len(steps), time.time() - t,
        u1.function_space().size() + p1.function_space().N + th.function_space().N))
