# Development: the PRODUCT SpMV kernel (flow_operator_apply) on an HBM-resident
# 10 M-row 7-point banded matrix (1.04 GB: does not fit the Infinity Cache) and
# on a 40 M-row one; SURVEY 8(d) asks for this beside the 114 MB pressure matrix.
import ctypes, os, sys
import numpy, torch
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import _hip, device
from flow_amd.fem.multigrid import CsrOperator


def banded(n, ny):
    offs = numpy.array([-ny - 1, -ny, -1, 0, 1, ny, ny + 1])
    i = numpy.arange(n)[:, None] + offs[None, :]
    ok = (i >= 0) & (i < n)
    rowptr = numpy.concatenate([[0], numpy.cumsum(ok.sum(axis=1))])
    vals = numpy.where(offs[None, :].repeat(n, 0)[ok] == 0, 6.0, -1.0)
    return sp.csr_matrix((vals, i[ok], rowptr), shape=(n, n))


lib = _hip.lib()
for n, ny in ((10000000, 1540), (40000000, 3080)):
    A = banded(n, ny)
    op = CsrOperator(A)
    x = torch.sin(torch.arange(n, dtype=torch.float64, device=device.get()))
    y = device.zeros(n)
    st = _hip.stream()
    args = (ctypes.byref(op.op), _hip.f64(x, n), _hip.f64(y, n), st)
    _hip.check(lib.flow_operator_apply(*args))
    err = abs(device.to_host(y).numpy() - A.dot(device.to_host(x).numpy())).max()
    for _ in range(5):
        lib.flow_operator_apply(*args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    device.synchronize()
    e0.record(torch.cuda.current_stream())
    for _ in range(50):
        lib.flow_operator_apply(*args)
    e1.record(torch.cuda.current_stream())
    device.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 50
    B = 12 * A.nnz + 4 * (n + 1) + 16 * n
    print('banded 7-pt, %d rows, band %d: %.1f MB, %.1f us, %.0f GB/s = %.1f %% of 8 TB/s, max err %.1e'
          % (n, ny, B / 1e6, t * 1e6, B / t / 1e9, 100 * B / t / 8e12, err), flush=True)
    del op, x, y, A
