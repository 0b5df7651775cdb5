# -*- coding: utf-8 -*-
'''How good is the start vector of the step-size controller's projection?
Relative distance start -> solution per settled step, old rule (the fields
extrapolated through 2-3 points) against the increments extrapolated with the
least-squares cubic.   python tools/projection_start_check.py [nx] [steps] [mu]'''
from __future__ import print_function
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 772
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.00565
    from flow_amd import karman, fem
    import flow_amd.navier_stokes as navsto
    ny = max(2, int(round(nx * 509.0 / 2182.0)))
    prob = karman.KarmanProblem(nx, ny, mu=mu)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    prob.settle()
    seen = []
    orig = fem.project_magnitude

    def spy(u, tol=1e-7, initial_guess=None):
        g = initial_guess.data.clone() if initial_guess is not None else None
        out = orig(u, tol=tol, initial_guess=initial_guess)
        if g is not None:
            d = float((g - out.data).norm() / out.data.norm())
        else:
            d = float('nan')
        seen.append((d, out.solve_info.iterations, out.solve_info.residual))
        return out
    fem.project_magnitude = spy
    for _ in range(steps):
        prob.step()
    for row in seen:
        print('start error %.3e  corrections %d  |z| %.3e' % row)


if __name__ == '__main__':
    main()
