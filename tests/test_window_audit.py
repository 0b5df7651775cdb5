# -*- coding: utf-8 -*-
'''
Index audit of the strip-sharded kernels (CPU; no GPU needed).

The sharded solvers of flow_amd/parallel.py hand the CSR-stream kernels their
input vector as a WINDOW [e0, e1) of the rank's strip, addressed by global row
through a base pointer shifted by -e0.  Round 2 lost a GPU run to a memory
access fault there (profiles/NOTES.md section 6, "The world-3 fault"): the kernels then
gathered x[col] for every index pair a lane held -- idle lanes hold column 0 --
and on rank 2 of 3 that address lay below the solver's work buffer.  An
out-of-window read that lands on a mapped page is silent, so no GPU test can
prove the class closed; this audit does it on the host: per world size, rank
and tile it recomputes the exact set of x indices every kernel variant
dereferences (tests/access_model.py restates the index arithmetic: alignment
slack, trailing odd element, idle lanes, empty tiles) from the REAL structures
flow_amd.parallel builds (row blocks of the owned rows, level-0 multigrid
operators, cell ranges, block ILU plans) and asserts that it lies inside the
window the kernel is given.
'''
import numpy
import pytest
import torch

from flow_amd import _hip, fem, parallel
from flow_amd.fem import ilu
from flow_amd.fem.multigrid import Multigrid
from flow_amd.fem.space import scalar_layout
from oracle import fem_oracle as orc

import access_model as am
import oracle_harness as H

WORLDS = [2, 3, 8]


@pytest.fixture
def host_structs(monkeypatch):
    '''Let the product's setup classes build their structs over HOST tensors:
    nothing is launched here, only the index tables are inspected.'''
    def ptr(t, dtype, numel=None, name='operand'):
        if t is None:
            return None
        assert t.dtype == dtype and t.is_contiguous()
        assert numel is None or t.numel() >= numel
        import ctypes
        return ctypes.c_void_p(t.data_ptr())
    monkeypatch.setattr(_hip, '_ptr', ptr)
    return None


def _meshes():
    return [
        ('staircase-120x30', fem.karman_channel(120, 30)),
        ('fitted-120x30', fem.karman_channel(120, 30, fitted=True)),
        ('fitted-200x47', fem.karman_channel(200, 47, fitted=True)),
        ]


def _host(t):
    return t.cpu().numpy() if isinstance(t, torch.Tensor) else numpy.asarray(t)


def test_the_model_sees_the_round2_fault():
    '''The defect as it was: Jacobi-CG on the P2 mass matrix of the 120 x 30
    channel (tests/test_parallel_gpu.py, the first sharded solve of the test
    that faulted), z = B r ext-compact at work + 4096 + me doubles, handed to
    spmv_stream_kernel as z - e0.  Idle lanes gathered x[0]: on rank 2 of 3
    that is 4472 bytes BELOW the work buffer -- the page the fault report
    names (a 2 MiB boundary minus 0x2000) when the buffer starts a segment --
    while with 2 ranks the same stray read still fell inside the buffer.'''
    mesh = fem.karman_channel(120, 30)
    lay = scalar_layout(mesh, 2)
    rowptr = lay.pattern('rowptr').astype(numpy.int64)
    cols = lay.pattern('cols')
    from flow_amd.fem.space import csr_stream_rowblocks
    offset = {}
    for world in (2, 3):
        rb = parallel.Strips(mesh, world).blocks(lay)
        for g in range(world):
            s = rb.struct(g)
            blocks = csr_stream_rowblocks(
                rowptr[s.r0:s.r1 + 1] - rowptr[s.r0]) + s.r0
            lo, _ = am.window(rowptr, cols, blocks, 'stream_r2a')
            me = s.e1 - s.e0
            # byte offset of the lowest address read, relative to `work`
            offset[(world, g)] = 8 * (_hip.REDUCE_WORK + me + lo - s.e0)
            # today's kernel stays inside the window
            lo, hi = am.window(rowptr, cols, blocks, 'stream')
            assert s.e0 <= lo and hi < s.e1
    assert offset[(3, 2)] == -4472
    assert -8192 <= offset[(3, 2)] < -4096          # the page below the buffer
    assert all(v >= 0 for k, v in offset.items() if k != (3, 2))


@pytest.mark.parametrize('world', WORLDS)
def test_operator_kernels_stay_inside_the_window(host_structs, world):
    '''Kinds 0 (spmv_stream), 2 (block2: the assembled Jacobian) and 4 (pair:
    the vector mass matrix) on the owned row blocks of every rank.'''
    for name, mesh in _meshes():
        st = parallel.Strips(mesh, world)
        for degree in (1, 2):
            lay = scalar_layout(mesh, degree)
            rowptr = lay.pattern('rowptr')
            cols = lay.pattern('cols')
            # (block2 / pair have no empty-tile branch: every row of a square
            # finite element pattern holds its diagonal)
            assert numpy.diff(rowptr).min() >= 1
            covered = 0
            for g in range(world):
                v = parallel.View(lay, st, g)
                covered += v.r1 - v.r0
                for variant in ('stream', 'block2', 'pair'):
                    blocks = _host(v.rowblocks if variant == 'stream'
                                   else v.rowblocks2)
                    assert blocks[0] == v.r0 and blocks[-1] == v.r1
                    lo, hi = am.window(rowptr, cols, blocks, variant)
                    assert v.e0 <= lo and hi < v.e1, \
                        (name, degree, world, g, variant, (lo, hi),
                         (v.e0, v.e1))
            assert covered == lay.N


@pytest.mark.parametrize('world', WORLDS)
def test_cell_kernels_stay_inside_the_window(world):
    '''Residual / Jacobian-action / right-hand-side kernels visit the rank's
    cell range and index strip-compact vectors with every dof of those
    cells.'''
    for name, mesh in _meshes():
        st = parallel.Strips(mesh, world)
        for degree in (1, 2):
            lay = scalar_layout(mesh, degree)
            rb = st.blocks(lay)
            for g in range(world):
                s = rb.struct(g)
                c0, c1 = st.cells[g]
                cd = lay.cell_dofs[c0:c1]
                assert s.e0 <= cd.min() and cd.max() < s.e1, (name, degree, g)
                # the gather phase of the assembly reads scratch entries of the
                # rank's cells only: every contribution to an owned row comes
                # from a cell in [c0, c1)
                owned = (lay.cell_dofs >= s.r0) & (lay.cell_dofs < s.r1)
                touching = numpy.nonzero(owned.any(axis=1))[0]
                assert c0 <= touching.min() and touching.max() < c1


@pytest.mark.parametrize('world', WORLDS)
def test_multigrid_level0_stays_inside_the_window(host_structs, world):
    '''The strip-sharded V-cycle: Ah0 reads the residual window, Rg (the
    restriction cut to the owned columns; rows with no owned column are EMPTY
    tiles) reads the owned part of t in local numbering, Ps0 reads the
    replicated coarse vector.'''
    class Fake(object):
        kind = 0

        def __init__(self, lay, M):
            self.layout, self.M = lay, M

        def to_scipy(self):
            return self.M

    for name, mesh in _meshes()[1:]:
        lay = scalar_layout(mesh, 1)
        S = H.oracle_space(mesh, 1)
        K = orc.stiffness_matrix(S).tocsr()
        isbc = mesh.points[:, 0] > mesh.points[:, 0].max() - 1e-12
        D = numpy.where(isbc, 0.0, 1.0)
        import scipy.sparse as sp
        Kbc = (sp.diags(D).dot(K).dot(sp.diags(D))
               + sp.diags(isbc.astype(float))).tocsr()
        mg = Multigrid(Fake(lay, Kbc), isbc, coarsest=200)
        assert mg.nlevels >= 3
        st = parallel.Strips(mesh, world)
        empty_tiles = 0
        for g in range(world):
            v = parallel.View(lay, st, g)
            ms = parallel.MgShard(mg, v)
            lvl = mg.levels[0]
            for op, blocks, (w0, w1) in (
                    (lvl['Ah'], ms._keep[0], (v.e0, v.e1)),
                    (lvl['Ps'], ms._keep[1], (0, mg.sizes[1])),
                    (ms.Rg, ms.Rg._rb, (0, v.r1 - v.r0))):
                rowptr = _host(op._rowptr)
                cols = _host(op._cols)
                rb = _host(blocks)
                lo_t, hi_t = am.tile_accesses(rowptr, cols, rb, 'stream')
                sel = lo_t <= hi_t
                empty_tiles += int((~sel).sum())
                assert sel.any()
                assert w0 <= lo_t[sel].min() and hi_t[sel].max() < w1, \
                    (name, world, g, (w0, w1))
        # (the audit has seen the empty-tile path of the kernel)
        assert world == 2 or empty_tiles > 0
        # the one-collective form: the up-sweep (Ps and Ah tiles over the SAME
        # row blocks) covers the owned rows plus one ghost layer, the residual
        # window reaches two layers out; C cut to the owned columns reads the
        # owned part of a vector in local numbering
        try:
            deep = st.deep_blocks(lay, 2)
        except parallel.StripsTooThin:
            continue                        # (strips too thin at this size)
        for g in range(world):
            v = parallel.View(lay, st, g)
            rows2 = deep.struct(g)
            z0, z1 = st.deep_ranges(lay, 2)[g][1]
            ms = parallel.MgShard(mg, v, rows2, (z0, z1))
            assert ms.struct.z_lo == z0 and ms.struct.z_hi == z1
            assert rows2.e0 <= z0 <= v.r0 and v.r1 <= z1 <= rows2.e1
            up = _host(ms._keep[-1])
            assert up[0] == z0 and up[-1] == z1
            for op, (w0, w1) in ((lvl['Ah'], (rows2.e0, rows2.e1)),
                                 (lvl['Ps'], (0, mg.sizes[1]))):
                lo_t, hi_t = am.tile_accesses(_host(op._rowptr),
                                              _host(op._cols), up, 'stream')
                sel = lo_t <= hi_t
                assert w0 <= lo_t[sel].min() and hi_t[sel].max() < w1, \
                    (name, world, g, 'up', (w0, w1))
            lo_t, hi_t = am.tile_accesses(
                _host(ms.Cg._rowptr), _host(ms.Cg._cols), _host(ms.Cg._rb),
                'stream')
            sel = lo_t <= hi_t
            assert 0 <= lo_t[sel].min() and hi_t[sel].max() < v.r1 - v.r0


@pytest.mark.parametrize('world', WORLDS)
def test_block_ilu_plans_are_local(host_structs, world):
    '''The block-Jacobi ILU(0) of a strip works in LOCAL numbering on its own
    sweep vector: every column of the factor pattern and of the sweep streams
    (pad entries: column 0) addresses a row of the block.'''
    mesh = fem.karman_channel(120, 30, fitted=True)
    lay = scalar_layout(mesh, 2)
    st = parallel.Strips(mesh, world)
    rb = st.blocks(lay)
    for g in range(world):
        s = rb.struct(g)
        plan = ilu.IluPlan(lay, rows=(s.r0, s.r1))
        n = s.r1 - s.r0
        assert plan.n == n
        for key in ('cols', 'l_cols', 'u_cols', 'old_of_new', 'new_of_old'):
            a = plan.host[key]
            assert a.min() >= 0 and a.max() < n, (g, key)
        # the factor reads its entries from the FULL value planes
        rowptr = lay.pattern('rowptr')
        src = plan.host['src_pos']
        assert src.min() >= rowptr[s.r0] and src.max() < rowptr[s.r1]


@pytest.mark.parametrize('world', WORLDS)
def test_block_pmg_levels_are_local(host_structs, world, monkeypatch):
    '''The rank-local two-level cycle of the strips (parallel.local_pmg, the
    quad-based tiles of pmg_kernels.hip): both packed levels in LOCAL
    numbering -- every column the tiles gather lies inside the block, couplings
    that leave it point at their own row with keep = 0 --, cols / vals
    readable three entries past nnz (the `+4` padding contract of quads loaded
    from a base aligned down to a multiple of four), and the transfer tables
    name rows of the block or the dummy coarse row n1 (a zero in the work
    buffer).'''
    from flow_amd.fem import pmg
    from flow_amd.fem.space import csr_stream_rowblocks
    monkeypatch.setattr(pmg, 'COLS16', False)      # (a kernel call: GPU only)
    mesh = fem.karman_channel(120, 30, fitted=True)
    lay2, lay1 = scalar_layout(mesh, 2), scalar_layout(mesh, 1)
    st = parallel.Strips(mesh, world)
    for g in range(world):
        s2, s1 = st.blocks(lay2).struct(g), st.blocks(lay1).struct(g)
        for lay, s in ((lay2, s2), (lay1, s1)):
            lvl = pmg._Level(lay, (s.r0, s.r1))
            h = lvl._host
            n, nnz = s.r1 - s.r0, lvl.nnz
            assert len(h['cols']) >= nnz + 4
            assert lvl.vals.numel() >= 2 * (nnz + 4)
            rb = csr_stream_rowblocks(h['rowptr'],
                                      nnz_per_block=_hip.PMG_NNZ_PER_BLOCK)
            lo, hi, last = am.quad_tile_accesses(h['rowptr'], h['cols'], rb)
            sel = lo <= hi
            assert sel.all()                       # no empty tiles in a block
            assert lo.min() >= 0 and hi.max() < n, (world, g, lay.degree)
            assert last.max() < nnz + 4
            # dropped couplings: value 0 (keep), column = the own row
            row_of = numpy.repeat(numpy.arange(n), numpy.diff(h['rowptr']))
            dropped = h['keep'] == 0
            assert (h['cols'][:nnz][dropped] == row_of[dropped]).all()
            assert dropped.any() == (world > 1)
        ends, rptr, rsrc = pmg.local_transfer_tables(
            lay2, (s2.r0, s2.r1), (s1.r0, s1.r1))
        n2, n1 = s2.r1 - s2.r0, s1.r1 - s1.r0
        assert ends.min() >= 0 and ends.max() <= n1       # n1: the dummy row
        # (an edge dof belongs to the rank of its LOWER vertex: only a partner
        # vertex further right can lie outside the block)
        assert (ends == n1).any() == (g < world - 1)
        assert rsrc.min() >= 0 and rsrc.max() < n2
        assert len(rptr) == n1 + 1 and rptr[-1] == len(rsrc)


def test_mass_solver_tiles_stay_inside_the_vector(host_structs):
    '''The fp16 stream of the mass solver (mass_kernels.hip) over the whole
    pattern: gathers inside [0, n), quads readable past nnz.'''
    from flow_amd.fem.space import csr_stream_rowblocks
    for name, mesh in _meshes()[1:]:
        for degree in (1, 2):
            lay = scalar_layout(mesh, degree)
            cols = numpy.concatenate([lay.pattern('cols'),
                                      numpy.zeros(4, dtype=numpy.int64)])
            rb = csr_stream_rowblocks(lay.pattern('rowptr'),
                                      nnz_per_block=_hip.PMG_NNZ_PER_BLOCK)
            lo, hi, last = am.quad_tile_accesses(lay.pattern('rowptr'), cols, rb)
            assert lo.min() >= 0 and hi.max() < lay.N
            assert last.max() < lay.nnz + 4
            # the packed streams (flow_mass.packed16, flow_pmg_level.cols16 /
            # .packed on the same row blocks): every column offset from the
            # tile's lowest column fits in 16 bits, base + offset IS the column
            cbase, off, fits = am.cols16_tables(lay.pattern('rowptr'),
                                                lay.pattern('cols'), rb)
            assert fits, (name, degree, int(off.max()))
            tile = numpy.repeat(numpy.arange(len(rb) - 1), numpy.diff(
                lay.pattern('rowptr').astype(numpy.int64)[rb]))
            assert numpy.array_equal(cbase[tile] + off, lay.pattern('cols'))
