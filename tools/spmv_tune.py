# Development harness: times the SpMV kernel variants of tools/spmv_variants.hip
# on (a) the pressure matrix of the headline workload, (b) a scalar P2 mass
# matrix, (c) an HBM-resident 10 M-row 7-point banded matrix.
import ctypes, os, sys, time
import numpy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flow_amd import fem, device
from flow_amd.fem.space import csr_stream_rowblocks

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'libspmv_variants.so'))
lib.spmv_variant.restype = ctypes.c_int
lib.spmv_variant.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 7


def bench(name, rowptr, cols, vals, variants=(11, 15, 18, 19, 20, 21, 22, 23)):
    n = len(rowptr) - 1
    nnz = len(cols)
    dev = device.get()
    pad = lambda a: numpy.concatenate([a, numpy.zeros(4, dtype=a.dtype)])
    d_rowptr = torch.from_numpy(rowptr.astype(numpy.int32)).to(dev)
    d_cols = torch.from_numpy(pad(cols.astype(numpy.int32))).to(dev)
    d_vals = torch.from_numpy(pad(vals.astype(numpy.float64))).to(dev)
    x = torch.sin(torch.arange(n, dtype=torch.float64, device=dev))
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    rb256 = torch.from_numpy(csr_stream_rowblocks(rowptr)).to(dev)
    rb512 = torch.from_numpy(csr_stream_rowblocks(rowptr, 512, 4096)).to(dev)
    rbs = {
        2: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 2046)).to(dev),
        6: torch.from_numpy(csr_stream_rowblocks(rowptr, 512, 4094)).to(dev),
        7: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 2046)).to(dev),
        8: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 2046)).to(dev),
        9: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 4094)).to(dev),
        10: torch.from_numpy(csr_stream_rowblocks(rowptr, 128, 1022)).to(dev),
        }
    rbs.update({11: rbs[8], 12: rbs[6], 13: rbs[9], 14: rbs[10], 16: rbs[8],
                17: rbs[8],
                15: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 1022)).to(dev)})
    rbs.update({18: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 1534)).to(dev),
                19: torch.from_numpy(csr_stream_rowblocks(rowptr, 512, 2046)).to(dev),
                20: rbs[15], 23: rbs[15],
                21: torch.from_numpy(csr_stream_rowblocks(rowptr, 128, 510)).to(dev),
                22: torch.from_numpy(csr_stream_rowblocks(rowptr, 256, 510)).to(dev)})
    import scipy.sparse as sp
    ref = sp.csr_matrix((vals, cols, rowptr), shape=(n, n)).dot(x.cpu().numpy())
    B = 12 * nnz + 4 * (n + 1) + 16 * n
    st = torch.cuda.current_stream().cuda_stream
    print('%s: n=%d nnz=%d bytes=%.1f MB' % (name, n, nnz, B / 1e6))
    for v in variants:
        rb = rbs.get(v, rb512 if v == 4 else rb256)
        nb = rb.numel() - 1
        args = (v, n, nb, d_rowptr.data_ptr(), d_cols.data_ptr(), d_vals.data_ptr(),
                rb.data_ptr(), x.data_ptr(), y.data_ptr(), st)
        y.zero_()
        assert lib.spmv_variant(*args) == 0
        torch.cuda.synchronize()
        err = abs(y.cpu().numpy() - ref).max() / abs(ref).max()
        for _ in range(10):
            lib.spmv_variant(*args)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(100):
            lib.spmv_variant(*args)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 100
        print('  v%d: %.2f us  %.0f GB/s (%.1f%% of 8 TB/s)  err %.1e' % (
            v, t * 1e6, B / t / 1e9, 100 * B / t / 8e12, err))


def banded(n, ny):
    offs = numpy.array([-ny - 1, -ny, -1, 0, 1, ny, ny + 1])
    i = numpy.arange(n)[:, None] + offs[None, :]
    ok = (i >= 0) & (i < n)
    rowptr = numpy.concatenate([[0], numpy.cumsum(ok.sum(axis=1))])
    cols = i[ok]
    vals = numpy.where(offs[None, :].repeat(n, 0)[ok] == 0, 6.0, -1.0)
    return rowptr, cols, vals


if __name__ == '__main__':
    which = sys.argv[1:] or ['pressure', 'mass', 'stress']
    if 'pressure' in which:
        mesh = fem.karman_channel(2182, 509)
        lay = fem.FunctionSpace(mesh, 'CG', 1).layout
        rng = numpy.random.RandomState(0)
        bench('pressure P1 2182x509', lay.pattern('rowptr'), lay.pattern('cols'),
              rng.standard_normal(lay.nnz))
    if 'mass' in which:
        mesh = fem.karman_channel(2182, 509)
        lay = fem.FunctionSpace(mesh, 'CG', 2).layout
        rng = numpy.random.RandomState(0)
        bench('scalar P2 2182x509', lay.pattern('rowptr'), lay.pattern('cols'),
              rng.standard_normal(lay.nnz))
    if 'stress' in which:
        rowptr, cols, vals = banded(10000000, 1540)
        bench('banded 7-pt 10M rows (HBM resident)', rowptr, cols, vals)
