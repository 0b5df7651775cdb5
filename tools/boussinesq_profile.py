# -*- coding: utf-8 -*-
'''BASELINE config 4 in steady stepping: host profile (cProfile, cumulative) of
the coupled steps 8-11 (dt on its cap, p-multigrid in both solvers).
  python tools/boussinesq_profile.py [nx]'''
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    from flow_amd import fem, boussinesq, device
    mesh = fem.heater_box(nx, fitted=True)
    stepper = boussinesq.FixedPointStepper(boussinesq.HeaterBox(mesh), 1.0e-2)
    for _ in range(7):
        stepper.advance()
    device.synchronize()
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    for _ in range(4):
        stepper.advance()
    device.synchronize()
    pr.disable()
    print('4 coupled steps: %.1f ms each' % (250.0 * (time.time() - t)))
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
    pstats.Stats(pr).sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
