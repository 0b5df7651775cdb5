# -*- coding: utf-8 -*-
'''
flow_amd: MI355X-native implementation of nschloe/flow's Navier-Stokes
pressure-correction time step and heat operator (reference package layout:
flow/__init__.py:3-5: message, navier_stokes, stokes).  `heat` and
`stabilization` are imported explicitly by callers (`from flow_amd import
heat`), as in the reference.
'''
from . import message                                           # noqa: F401
from . import navier_stokes                                     # noqa: F401
from . import stokes                                            # noqa: F401

__all__ = ['message', 'navier_stokes', 'stokes']
