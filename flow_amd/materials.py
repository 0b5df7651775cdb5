# -*- coding: utf-8 -*-
'''
Temperature-dependent properties of liquid water (T in kelvin).  Stands in for
the third-party `materials.water.*` the reference's drivers call
(tests/test_boussinesq.py:106-110, tests/test_karman_vortex_street.py:183),
which is not available offline.  Written with plain arithmetic so the functions
accept floats, numpy arrays and torch tensors alike (fem.NodalExpression
evaluates them on the device).  Standard correlations, not the (unknown)
formulas of that package: results of the Boussinesq driver are therefore
comparable with the reference qualitatively only (parity unpinned).
'''


def density(T):
    '''Kell (1975), kg/m^3, 273 K .. 423 K.'''
    t = T - 273.15
    num = (999.83952 + t * (16.945176 + t * (-7.9870401e-3 + t * (
        -46.170461e-6 + t * (105.56302e-9 - 280.54253e-12 * t)))))
    return num / (1.0 + 16.879850e-3 * t)


def dynamic_viscosity(T):
    '''Vogel-type fit, Pa s.'''
    return 2.414e-5 * 10.0**(247.8 / (T - 140.0))


def specific_heat_capacity(T):
    '''J/(kg K); nearly constant between 280 K and 340 K.'''
    t = T - 273.15
    return 4217.4 - 3.720283 * t + 0.1412855 * t**2 - 2.654387e-3 * t**3 \
        + 2.093236e-5 * t**4


def thermal_conductivity(T):
    '''W/(m K).'''
    return -0.5752 + 6.397e-3 * T - 8.151e-6 * T * T
