import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flow_amd import karman
import flow_amd.navier_stokes as navsto
for auto in (True, False):
    navsto.solver_parameters['newton']['pmg']['coarse_auto']=auto
    prob = karman.KarmanProblem(160, 37, mu=0.02)
    prob.set_initial_profile()
    prob.dt = prob.hmax / 0.016
    for k in range(2):
        i = prob.step(adapt=False)
        print(auto, k, i.get('pmg_coarse_steps'), i.get('pmg_contraction'), i['newton_linear_applications'], i['newton_preconditioner'], ['%.1e'%r for r in i['newton_residuals']], i.get('newton_linear_residuals'))
    print('rejected' , 'pmg_rejected' in prob.W.layout._dev)
