# -*- coding: utf-8 -*-
'''Pressure-correction schemes (Chorin, IPCS, Rotational) and their solver
settings; the names the reference exports from flow.navier_stokes.'''
from .pressure_correction import (                               # noqa: F401
    Chorin, IPCS, Rotational, solver_parameters, last_step_info, set_mode,
    forget_history,
    )

__all__ = ['Chorin', 'IPCS', 'Rotational', 'solver_parameters',
           'last_step_info', 'set_mode', 'forget_history']
