# -*- coding: utf-8 -*-
'''
Minimal finite-element front end: the vocabulary the reference's drivers import
from dolfin (tests/test_navier_stokes.py:10-14, tests/test_karman_vortex_street.py:7-11,
tests/test_boussinesq.py:14-18, tests/test_sealed_box.py:9-13), restricted to
what the Navier-Stokes / heat hot path needs.  dolfin is not available on the
GPU box, so the counterpart drivers import these names from here instead.
'''
from .mesh import (                                             # noqa: F401
    Mesh, Point, RectangleMesh, UnitSquareMesh, rectangle_with_hole,
    karman_channel, karman_channel_graded, heater_box, heater_box_coarse,
    )
from .space import (                                            # noqa: F401
    FunctionSpace, VectorFunctionSpace, FiniteElement, VectorElement,
    )
from .function import (                                         # noqa: F401
    Function, Constant, Expression, NodalExpression, Vector,
    as_cell_coefficient, cell_lattice_points, scalar_value,
    )
from .bcs import DirichletBC, SubDomain                         # noqa: F401
from .io import XDMFFile, mpi_comm_world, read_mesh             # noqa: F401
from .space import MixedFunctionSpace                           # noqa: F401

DOLFIN_EPS = 3.0e-16
triangle = 'triangle'
pi = 3.141592653589793


def __getattr__(name):
    # the operations below run on the HIP path; import them lazily so that the
    # host-only parts (meshes, spaces, BC search) work without the library
    if name in ('project', 'interpolate', 'errornorm', 'norm', 'assemble_mass',
                'assemble_stiffness', 'integral', 'project_magnitude', 'ops'):
        import importlib
        ops = importlib.import_module('.ops', __name__)
        return ops if name == 'ops' else getattr(ops, name)
    raise AttributeError(name)
