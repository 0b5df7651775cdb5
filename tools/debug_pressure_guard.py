# -*- coding: utf-8 -*-
'''Development: |B b| against |B (b - A p0)| of the pressure solve on a settled
small Karman problem -- is the plain start p0 really worse than zero?'''
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy
import ctypes
from flow_amd import karman, device, _hip, fem
from flow_amd.fem import ops
import flow_amd.navier_stokes as navsto
from flow_amd.navier_stokes import pressure_correction as pc

navsto.solver_parameters['pressure']['mg_coarsest'] = 200
prob = karman.KarmanProblem(193, 45, mu=0.0226)
prob.prepare(); prob.reset(1e-5); prob.set_initial_stokes(); navsto.set_mode('parity')
prob.settle()
orig = pc._pressure_cg
def spy(A, dinv, prec, b, x, tol, par, fallback=False):
    coarse, mg = prec
    n = A.size
    lib = _hip.lib()
    def B(v):
        z = device.empty(n)
        _hip.check(lib.flow_mg_apply(ctypes.byref(mg.struct), n, _hip.f64(v), _hip.f64(z), _hip.stream()))
        return z
    Bb = B(b)
    w = device.empty(n); A.apply(x, w)
    r = _hip.clone(b); ops.axpby(-1.0, w, 1.0, r)
    Br = B(r)
    print('|b| %.3e |r0| %.3e  |Bb| %.3e |Br0| %.3e  |x0| %.3e' % (
        ops.vector_norm(b), ops.vector_norm(r), ops.vector_norm(Bb), ops.vector_norm(Br), ops.vector_norm(x)))
    sol = orig(A, dinv, prec, b, x, tol, par, fallback)
    print('   ->', sol, 'dropped', sol.starts_dropped, '|x| %.3e' % ops.vector_norm(x))
    return sol
pc._pressure_cg = spy
for mode in ('zero', 'extrapolated'):
    navsto.solver_parameters['pressure']['start'] = mode
    print(mode)
    for k in range(8):
        prob.step()
