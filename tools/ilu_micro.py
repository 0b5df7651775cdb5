# Development: time of one ILU(0) application (fp64 and packed fp32 streams) on
# the headline workload's velocity Jacobian pattern (two P2 blocks).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from flow_amd import fem, device
from flow_amd.fem import ops, ilu
nx = int(os.environ.get('NX', 2182))
mesh = fem.karman_channel(nx, None) if nx != 2182 else fem.karman_channel(2182, 509)
W = fem.VectorFunctionSpace(mesh, 'CG', 2)
V = W.collapse()
lay = V.layout
M = ops.assemble_mass(V)
K = ops.assemble_stiffness(V)
vals = M.vals / 0.02 + 0.002 * K.vals
A = ops.Matrix(lay, 1, torch.cat([vals, vals]))
n = 2 * lay.N
r = device.to_device(numpy.random.RandomState(0).standard_normal(n))
z = device.zeros(n)
for packed in (False, True):
    pre = ilu.Ilu0(A, packed=packed)
    plan = pre.plan
    if not packed:
        w = numpy.diff(plan.host['l_sl_off']) // 64
        print('colours %d, slices %d, L widths: mean %.1f max %d; fill L %.2f U %.2f'
              % (plan.ncolours, len(w), w.mean(), w.max(), plan.fill_l, plan.fill_u))
    for _ in range(5):
        pre.solve(r, z)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    reps = 50
    e0.record(s)
    for _ in range(reps):
        pre.solve(r, z)
    e1.record(s)
    torch.cuda.synchronize()
    print('packed=%s batch=%s: %.1f us per application' % (
        packed, os.environ.get('FLOW_ILU_BATCH', '12'), e0.elapsed_time(e1) * 1e3 / reps), flush=True)
    del pre
