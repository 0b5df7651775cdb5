# -*- coding: utf-8 -*-
'''Where does the HOST spend a time step?  cProfile over plateau steps of the
proxy-size Karman run (the GPU is launch-/latency-bound there: host time
between launches is idle GPU time).
  python tools/host_profile.py [nx] [steps] [mu]'''
from __future__ import print_function
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 772
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    mu = float(sys.argv[3]) if len(sys.argv) > 3 else 0.00565
    from flow_amd import karman
    import flow_amd.navier_stokes as navsto
    ny = max(2, int(round(nx * 509.0 / 2182.0)))
    prob = karman.KarmanProblem(nx, ny, mu=mu)
    prob.prepare()
    prob.reset(1.0e-5)
    prob.set_initial_stokes()
    navsto.set_mode('parity')
    prob.settle()
    for _ in range(10):
        prob.step()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        prob.step()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(22)


if __name__ == '__main__':
    main()
