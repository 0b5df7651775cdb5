# -*- coding: utf-8 -*-
'''TEST INFRASTRUCTURE: ctypes loader of oracle/cpu_cg.c (see its header).'''
import ctypes
import os
import subprocess

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(native=False, out_dir=None):
    '''Compile cpu_cg.c; native=True adds -march=native (cpu_baseline leg).'''
    out_dir = out_dir or _HERE
    out = os.path.join(out_dir, 'liboracle_cpu%s.so' % ('_native' if native else ''))
    cmd = ['gcc', '-O3', '-fopenmp', '-fPIC', '-shared', '-std=c99']
    if native:
        cmd.append('-march=native')
    cmd += [os.path.join(_HERE, 'cpu_cg.c'), '-o', out, '-lm']
    subprocess.check_call(cmd)
    return out


def usable_cores():
    '''CPUs this process can really use: the affinity mask capped by the
    cgroup CPU quota (a GPU box hands one tenant a share of its cores;
    omp_get_max_threads() sees all of them and oversubscribes).'''
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if quota > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return n


def load(path=None):
    path = path or os.path.join(_HERE, 'liboracle_cpu.so')
    if not os.path.isfile(path):
        path = build()
    lib = ctypes.CDLL(path)
    ip = numpy.ctypeslib.ndpointer(numpy.int32, flags='C_CONTIGUOUS')
    dp = numpy.ctypeslib.ndpointer(numpy.float64, flags='C_CONTIGUOUS')
    lib.oracle_num_threads.restype = ctypes.c_int
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads.argtypes = [ctypes.c_int]
    if not os.environ.get('OMP_NUM_THREADS'):
        lib.oracle_set_threads(usable_cores())
    lib.oracle_spmv_csr.restype = None
    lib.oracle_spmv_csr.argtypes = [ctypes.c_int, ip, ip, dp, dp, dp]
    lib.oracle_jacobi_cg.restype = ctypes.c_int
    lib.oracle_jacobi_cg.argtypes = [
        ctypes.c_int, ip, ip, dp, dp, dp, dp, ctypes.c_double, ctypes.c_double,
        ctypes.c_int, dp, ctypes.POINTER(ctypes.c_int),
        ctypes.POINTER(ctypes.c_double),
        ]
    return lib


def jacobi_cg(lib, A, b, rtol, atol=0.0, maxit=1000, x0=None):
    '''A: scipy CSR.  Returns (x, iterations, residual, converged).'''
    n = A.shape[0]
    x = numpy.zeros(n) if x0 is None else numpy.array(x0, dtype=float)
    dinv = numpy.ascontiguousarray(1.0 / A.diagonal())
    work = numpy.empty(4 * n)
    its = ctypes.c_int(0)
    res = ctypes.c_double(0.0)
    rc = lib.oracle_jacobi_cg(
        n, A.indptr.astype(numpy.int32), A.indices.astype(numpy.int32),
        numpy.ascontiguousarray(A.data, dtype=float), dinv,
        numpy.ascontiguousarray(b, dtype=float), x, rtol, atol, maxit, work,
        ctypes.byref(its), ctypes.byref(res)
        )
    return x, its.value, res.value, rc == 0


# -- like-for-like pressure solve: multigrid-preconditioned CG -----------------
_VP = ctypes.c_void_p


def _proto_mg(lib):
    if getattr(lib, '_mg_proto', False):
        return
    ip = numpy.ctypeslib.ndpointer(numpy.int32, flags='C_CONTIGUOUS')
    dp = numpy.ctypeslib.ndpointer(numpy.float64, flags='C_CONTIGUOUS')
    lib.oracle_csr_create.restype = _VP
    lib.oracle_csr_create.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, dp]
    lib.oracle_csr_destroy.restype = None
    lib.oracle_csr_destroy.argtypes = [_VP]
    lib.oracle_csr_spmv.restype = None
    lib.oracle_csr_spmv.argtypes = [_VP, _VP, _VP]
    lib.oracle_vec_create.restype = _VP
    lib.oracle_vec_create.argtypes = [ctypes.c_int, _VP]
    lib.oracle_vec_read.restype = None
    lib.oracle_vec_read.argtypes = [ctypes.c_int, _VP, dp]
    lib.oracle_vec_destroy.restype = None
    lib.oracle_vec_destroy.argtypes = [_VP]
    lib.oracle_mg_cg.restype = ctypes.c_int
    lib.oracle_mg_cg.argtypes = [
        ctypes.c_int, _VP, _VP, _VP, _VP, ctypes.c_int, dp, ctypes.c_double,
        _VP, _VP, ctypes.c_double, ctypes.c_double, ctypes.c_int,
        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
        ]
    lib._mg_proto = True


class Csr(object):
    '''A scipy CSR matrix copied into first-touched library memory.'''

    def __init__(self, lib, M):
        _proto_mg(lib)
        M = M.tocsr()
        M.sort_indices()
        self.lib = lib
        self.shape = M.shape
        self.nnz = M.nnz
        self.h = lib.oracle_csr_create(
            M.shape[0], M.shape[1], M.indptr.astype(numpy.int32),
            M.indices.astype(numpy.int32),
            numpy.ascontiguousarray(M.data, dtype=float))

    def __del__(self):
        if getattr(self, 'h', None):
            self.lib.oracle_csr_destroy(self.h)
            self.h = None


class Vec(object):
    def __init__(self, lib, arr):
        _proto_mg(lib)
        arr = numpy.ascontiguousarray(arr, dtype=float)
        self.lib = lib
        self.n = len(arr)
        self.h = lib.oracle_vec_create(self.n, arr.ctypes.data_as(_VP))

    def get(self):
        out = numpy.empty(self.n)
        self.lib.oracle_vec_read(self.n, self.h, out)
        return out

    def __del__(self):
        if getattr(self, 'h', None):
            self.lib.oracle_vec_destroy(self.h)
            self.h = None


class MgHierarchy(object):
    '''levels: [(A_l, D_l, P_l)] scipy CSR / diagonal / CSR per level (finest
    first), Ainv: dense (pseudo-)inverse of the coarsest Galerkin operator,
    omega: Jacobi damping -- the same objects the product's V-cycle is built
    from (flow_amd/fem/multigrid.py), handed in as data.'''

    def __init__(self, lib, levels, Ainv, omega):
        _proto_mg(lib)
        self.lib = lib
        self.A = [Csr(lib, A) for A, _, _ in levels]
        self.P = [Csr(lib, P) for _, _, P in levels]
        self.R = [Csr(lib, P.T.tocsr()) for _, _, P in levels]
        self.dinv = [Vec(lib, 1.0 / D) for _, D, _ in levels]
        self.Ainv = numpy.ascontiguousarray(Ainv, dtype=float)
        self.omega = float(omega)
        self.n = levels[0][0].shape[0] if levels else self.Ainv.shape[0]

    def _arr(self, objs):
        return (_VP * max(len(objs), 1))(*[o.h for o in objs])

    def cg(self, b, rtol, atol=0.0, maxit=1000, x0=None):
        '''Returns (x, iterations, preconditioned residual, converged).'''
        bv = Vec(self.lib, b)
        xv = Vec(self.lib, numpy.zeros(self.n) if x0 is None else x0)
        its = ctypes.c_int(0)
        res = ctypes.c_double(0.0)
        rc = self.lib.oracle_mg_cg(
            len(self.A) + 1, self._arr(self.A), self._arr(self.P),
            self._arr(self.R), self._arr(self.dinv), self.Ainv.shape[0],
            self.Ainv, self.omega, bv.h, xv.h, rtol, atol, maxit,
            ctypes.byref(its), ctypes.byref(res))
        return xv.get(), its.value, res.value, rc == 0
