# -*- coding: utf-8 -*-
'''A loaded, graded, unstructured Karman mesh on the fast path: the Delaunay
channel of fem.karman_channel_graded (the stand-in for the reference's gmsh
mesh, tests/test_karman_vortex_street.py:26-53) written to MSH, read back
(renumbered along the channel: Mesh.reordered), stepped like the bench's
workload -- Stokes start, settle, a window of steps -- next to the structured
body-fitted channel of the same DoF count.
  python tools/graded_mesh_time.py [lcar] [steps]       (3.2e-4: ~1 M DoF)
REORDER=0: the file's numbering as it is (what the reordering is worth).'''
from __future__ import print_function
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(prob, label, steps):
    from flow_amd import device, _hip
    import flow_amd.navier_stokes as navsto
    t0 = time.time()
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    n = prob.settle()
    device.synchronize()
    setup = time.time() - t0
    for _ in range(8):
        prob.step()
    device.synchronize()
    t1 = time.time()
    l0 = _hip.launch_count()
    infos = [prob.step() for _ in range(steps)]
    device.synchronize()
    launches = _hip.launch_count() - l0
    ms = 1e3 * (time.time() - t1) / steps
    print('%-28s %7d DoF  hmin %.2e hmax %.2e  dt %.3e | %6.2f ms/step | '
          'Newton %.2f its, GMRES %.1f (%s), pressure %.1f, corrections %.1f | '
          'setup %.1f s (%d settle steps) | launches/step %.0f, contraction '
          'cycle / smoother %s'
          % (label, prob.num_dofs(), prob.mesh.hmin(), prob.mesh.hmax(), prob.dt,
             ms, sum(len(i['newton_residuals']) - 1 for i in infos) / float(steps),
             sum(sum(i['newton_linear_applications']) for i in infos) / float(steps),
             infos[-1].get('newton_preconditioner'),
             sum(i['pressure'].iterations for i in infos) / float(steps),
             sum(i['correction'].iterations for i in infos) / float(steps),
             setup, n, launches / float(steps),
             infos[-1].get('tl_contraction', prob.W.layout._dev.get(
                 'jacobian_tl') and (prob.W.layout._dev['jacobian_tl'].contraction,
                                     prob.W.layout._dev['jacobian_tl'].contraction_bare))),
          flush=True)


def main():
    # FALLBACK=ilu0|tlilu, TLILU=pre,post,coarse_sweeps: what replaces a
    # rejected Chebyshev cycle (solver_parameters['newton'])
    import flow_amd.navier_stokes as navsto
    npar = navsto.solver_parameters['newton']
    if os.environ.get('FALLBACK'):
        npar['fallback'] = os.environ['FALLBACK']
    if os.environ.get('TLILU'):
        a, b, c = [int(x) for x in os.environ['TLILU'].split(',')]
        npar['tlilu'] = {'pre': a, 'post': b, 'coarse_sweeps': c}
    print('fallback %s %r' % (npar['fallback'], npar['tlilu']), flush=True)
    lcar = float(sys.argv[1]) if len(sys.argv) > 1 else 3.2e-4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from flow_amd import fem, karman
    from flow_amd.fem import io
    t0 = time.time()
    # FAR=<ratio>: lcar_far = ratio * lcar (default 4: graded; 1: the ONE
    # characteristic length the reference's driver hands gmsh,
    # tests/test_karman_vortex_street.py:35-45 -- quasi-uniform)
    far = float(os.environ.get('FAR', '4'))
    m = fem.karman_channel_graded(lcar, lcar_far=far * lcar)
    path = '/tmp/karman_graded_%g_%g.msh' % (lcar, far)
    io.write_msh(path, m, binary=True)
    print('generated %d vertices in %.1f s, written to %s' % (
        m.num_vertices(), time.time() - t0, path), flush=True)
    if os.environ.get('REORDER', '1') == '1':
        mesh = fem.Mesh(path)
        label = 'graded Delaunay, reordered' if far != 1.0 \
            else 'uniform Delaunay, reordered'
    else:
        mesh = io.read_mesh(path, reorder=False)
        label = 'graded Delaunay, file order'
    prob = karman.KarmanProblem(mesh=mesh) if label.endswith('reordered') \
        else karman.KarmanProblem.__new__(karman.KarmanProblem)
    if not label.endswith('reordered'):
        # (bypass the automatic reordering of KarmanProblem)
        import types
        mesh.bandwidth = types.MethodType(lambda self: 0, mesh)
        prob.__init__(mesh=mesh)
    run(prob, label, steps)
    if os.environ.get('NO_STRUCTURED'):
        return
    # the structured channel with as many DoF
    nv = mesh.num_vertices()
    nx = int(round((nv * 0.6 / 0.14)**0.5))
    run(karman.KarmanProblem(nx), 'structured body-fitted %d' % nx, steps)


if __name__ == '__main__':
    main()
