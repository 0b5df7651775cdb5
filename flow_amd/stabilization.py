# -*- coding: utf-8 -*-
#
'''
Stabilization techniques for PDEs with dominating convection: the SUPG
parameter tau of flow/stabilization.py (reference :13-152).

The reference JIT-compiles a C++ `Expression` (`SupgStab::eval`, :50-143) that
dolfin evaluates at the three vertices of every cell (`degree=1`, :147).  Here
tau is a device function of the heat assembly kernel
(flow_amd/csrc/assembly_kernels.hip, `supg_tau`), evaluated at the same points:

    tau = h^2 / (4 eps p) * xi(Pe),   Pe = |b| h / (2 p eps),
    h   = 4 |b| area / sum_edges |e_y b_x - e_x b_y|      (directed diameter),
    xi  = (coth Pe - 1/Pe)/Pe  for Pe > 1e-5, else 1/3 - Pe^2/45 + 2 Pe^4/945,
    tau = 0 for |b| < 1e-10;  tau > 1e3 is an error (the reference throws).
'''
import ctypes

import numpy
import torch

from .fem import ops
from . import _hip
from . import device


class SupgTau(object):
    '''Handle returned by `supg()`; mirrors the attributes the reference sets on
    its Expression (convection, mesh, epsilon, p; reference :148-151).'''

    def __init__(self, mesh, convection, epsilon, p):
        self.mesh = mesh
        self.convection = convection
        self.epsilon = float(epsilon)
        self.p = int(p)
        self.degree = 1

    def cell_vertex_values(self):
        '''tau at the three vertices of every cell, (Nc, 3) numpy array,
        computed by the HIP kernel (K14).'''
        from .fem.space import scalar_layout
        lib = _hip.lib()
        mesh = self.mesh
        W = self.convection.function_space()
        Q = scalar_layout(mesh, self.p)
        nc = mesh.num_cells()
        buf = ops.scratch(mesh, 2 * Q.nloc**2 * nc)
        A = device.empty(Q.nnz)
        Ms = device.empty(Q.nnz)
        tau = device.empty(3 * nc)
        status = device.zeros(1, dtype=torch.int32)
        _hip.check(lib.flow_assemble_heat(
            ctypes.byref(ops.mesh_struct(mesh)),
            ctypes.byref(ops.space_struct(Q)),
            ctypes.byref(ops.space_struct(W.layout)),
            _hip.f64(self.convection.data, W.size()), self.epsilon, 1.0, 1,
            _hip.f64(buf), _hip.f64(A), _hip.f64(Ms), _hip.f64(tau),
            _hip.i32(status), _hip.stream()
            ))
        if int(device.to_host(status).item()) != 0:
            raise RuntimeError('SUPG: tau > 1e3')
        return numpy.ascontiguousarray(device.to_host(tau).numpy().reshape(3, nc).T)


def supg(mesh, convection, diffusion, element_degree):
    '''For each cell, tau as in (3) of the SOLD review cited by the reference
    (:14-36).  `convection` is a vector Function, `diffusion` a number.'''
    return SupgTau(mesh, convection, diffusion, element_degree)
