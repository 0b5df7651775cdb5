# -*- coding: utf-8 -*-
'''Development: the 12-step trajectory of tests/test_large_parity.py (53 k DoF,
cell Peclet ~12) with the preconditioner's verdicts printed.'''
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy
import large_cases
import flow_amd.navier_stokes as navsto
npar = navsto.solver_parameters['newton']
for kv in sys.argv[1:]:
    k, v = kv.split('=')
    npar[k] = type(npar[k])(v) if not isinstance(npar[k], dict) else eval(v)
npar['linear_maxit'] = 600
case = large_cases.KarmanStepCase(160, 37)
navsto.forget_history(case.W)
up, pp = case.u0, case.p0
for k in range(12):
    try:
        up, pp, _ = case.product_step(u0=up, p0=pp)
    except Exception as e:
        print('step', k, 'FAILED', e)
        i = navsto.last_step_info
        print('   ', i.get('newton_preconditioner'), i.get('tl_contraction'), i.get('pmg_contraction'))
        break
    i = navsto.last_step_info
    print('step', k, i['newton_preconditioner'], i.get('tl_contraction'),
          i['newton_linear_applications'], ['%.1e' % r for r in i['newton_residuals']], flush=True)
