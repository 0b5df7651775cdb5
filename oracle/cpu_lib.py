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


def load(path=None):
    path = path or os.path.join(_HERE, 'liboracle_cpu.so')
    if not os.path.isfile(path):
        path = build()
    lib = ctypes.CDLL(path)
    ip = numpy.ctypeslib.ndpointer(numpy.int32, flags='C_CONTIGUOUS')
    dp = numpy.ctypeslib.ndpointer(numpy.float64, flags='C_CONTIGUOUS')
    lib.oracle_num_threads.restype = ctypes.c_int
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
