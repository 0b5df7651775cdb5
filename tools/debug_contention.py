# Development: are the kernels bitwise reproducible when several processes share
# the GPU?  Every process runs the same sequence and prints checksums.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch, torch.multiprocessing as mp


def worker(rank, nproc, out):
    from flow_amd import fem, device
    from flow_amd.fem import ops, ilu
    from flow_amd.fem.mesh import RectangleMesh
    mesh = RectangleMesh((0.0, 0.0), (2.0, 1.0), 96, 48)
    V = fem.FunctionSpace(mesh, 'Lagrange', 2)
    lay = V.layout
    res = []
    for trial in range(6):
        M = ops.assemble_mass(V)
        K = ops.assemble_stiffness(V)
        A = ops.Matrix(lay, 1)
        for p in (0, 1):
            A.plane(p).copy_(M.vals[:lay.nnz] + 0.01 * K.vals[:lay.nnz])
        P = ilu.Ilu0(A)
        n2 = 2 * lay.N
        r = torch.sin(torch.arange(n2, dtype=torch.float64, device=device.get()))
        z = device.zeros(n2)
        P.solve(r, z)
        x = device.zeros(n2)
        sol = ops.krylov_solve('bicgstab', A, r, x, rtol=1e-6, maxit=100, ilu=P, check_every=2)
        y = device.zeros(n2)
        sol2 = ops.krylov_solve('cg', A, r, y, rtol=1e-10, maxit=1000)
        h = lambda t: device.to_host(t).numpy().tobytes().__hash__()
        # repeated applications: every result must equal the first
        z2 = device.zeros(n2)
        bad_ilu = 0
        for k in range(300):
            P.solve(r, z2)
            bad_ilu += int(not torch.equal(z2, z))
        w0 = device.zeros(n2); A.apply(r, w0)
        w1 = device.zeros(n2)
        bad_spmv = 0
        for k in range(300):
            A.apply(r, w1)
            bad_spmv += int(not torch.equal(w1, w0))
        xj = device.zeros(n2)
        solj = ops.krylov_solve('bicgstab', A, r, xj, rtol=1e-6, maxit=2000, check_every=2)
        xk = device.zeros(n2)
        solk = ops.krylov_solve('bicgstab', A, r, xk, rtol=1e-6, maxit=2000, check_every=1000)
        yk = device.zeros(n2)
        sol3 = ops.krylov_solve('cg', A, r, yk, rtol=1e-10, maxit=1000, check_every=2)
        res.append((h(A.vals), h(P.lu), h(z), sol.iterations, h(x), sol2.iterations, h(y),
                    bad_ilu, bad_spmv, solj.iterations, h(xj), h(xk), sol3.iterations, h(yk)))
    out[rank] = res


if __name__ == '__main__':
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    m = mp.get_context('spawn').Manager()
    out = m.dict()
    os.environ['PYTHONHASHSEED'] = '0'
    mp.spawn(worker, args=(nproc, out), nprocs=nproc, join=True)
    names = ('A', 'lu', 'ilu z', 'bicg its', 'bicg x', 'cg its', 'cg y', 'repeated ilu mismatches', 'repeated spmv mismatches', 'bicg-jacobi its', 'bicg-jacobi x', 'bicg-jacobi no checks x', 'cg check2 its', 'cg check2 y')
    ref = out[0][0]
    for r in range(nproc):
        for t, row in enumerate(out[r]):
            bad = [names[k] for k in range(len(names)) if row[k] != ref[k]]
            if bad:
                print('rank %d trial %d differs in %s' % (r, t, bad))
    print('done; reference', ref[3], ref[5], ref[7:10])
    for r in range(nproc):
        print('rank', r, [(row[3], row[7], row[8], row[9]) for row in out[r]])
