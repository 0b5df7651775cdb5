# -*- coding: utf-8 -*-
'''
Host-side model of WHICH entries of the input vector `x` every CSR-stream kernel
variant of flow_amd/csrc/la_kernels.hip dereferences, tile by tile (helper, not
a test).  The strip-sharded solvers hand these kernels `x` as a window
[e0, e1) of a vector, addressed by global row through a base pointer shifted
by -e0: an index outside the window is an address outside the allocation (a
GPU memory-access fault when it lands on an unmapped page, a silent read of
foreign memory when it does not).  The model restates the kernels' index
arithmetic -- tile base aligned down to an even nonzero, index PAIRS per
lane, the trailing odd element, idle lanes, empty tiles -- so that a CPU test
can prove the access set lies inside the window for any partition.

Variants (the kernel each one restates):
  'stream'       stream_tile_row_sum: spmv_stream_kernel, mg_level_kernel
  'block2'       spmv_stream_block2_kernel
  'pair'         spmv_stream_pair_kernel (both components use the same indices)
  'quad'         pmg_tile_row_sum (pmg_kernels.hip) / mass_tile_row_sum
                 (mass_kernels.hip): QUADS of nonzeros per lane from a base
                 aligned down to a multiple of four, FLOW_PMG_NNZ_PER_BLOCK
                 nonzeros per tile; arrays readable three entries past nnz
  'stream_r2a'   stream_tile_row_sum as it was when the world-3 fault of round 2
                 happened (every lane gathers x[col] of whatever index pair it
                 loaded; idle lanes hold column 0) -- kept to show that the
                 audit sees that defect
'''
import numpy

from flow_amd import _hip

BLOCK = 256                 # kBlock


def pairs_per_lane(variant):
    '''kPairs (kinds 0, 1: a build constant of the library) / kPairs2.'''
    kind = 2 if variant in ('block2', 'pair') else 0
    return (_hip.spmv_tile_nnz(kind) + 2) // (2 * BLOCK)


def tile_accesses(rowptr, cols, rowblocks, variant='stream'):
    '''(lo, hi): per tile the smallest and the largest index of x the kernel
    dereferences in its gather phase (lo > hi: none).  cols must be readable
    one pair past the last nonzero, as the device arrays are (padding 0).'''
    rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
    cols = numpy.asarray(cols, dtype=numpy.int64)
    rb = numpy.asarray(rowblocks, dtype=numpy.int64)
    nt = len(rb) - 1
    k0 = rowptr[rb[:-1]]
    k1 = rowptr[rb[1:]]
    ka = k0 & ~1
    lo_e = (k0 - ka)[:, None]
    hi_e = (k1 - ka)[:, None]
    npair = ((k1 - ka + 1) >> 1)[:, None]
    lanes = BLOCK * pairs_per_lane(variant)      # index pairs per tile
    assert (k1 - ka <= 2 * lanes).all(), 'row block larger than the tile'
    p = numpy.arange(lanes)[None, :]
    e = 2 * p
    pad = numpy.zeros(2 * lanes + 4, dtype=numpy.int64)
    colsp = numpy.concatenate([cols, pad])          # (never used out of range
    cx = colsp[ka[:, None] + e]                     #  by a kernel: see `ok`)
    cy = colsp[ka[:, None] + e + 1]
    ok = p < npair
    nonempty = (k0 < k1)[:, None]
    big = numpy.iinfo(numpy.int64).max
    if variant == 'stream':
        safe = colsp[numpy.where(k0 < k1, k0, numpy.maximum(k0 - 1, 0))][:, None]
        # idle lanes hold (0, 0) and select `safe` like the slack entries
        ix = numpy.where((e >= lo_e) & (e < hi_e), numpy.where(ok, cx, 0), safe)
        iy = numpy.where(e + 1 < hi_e, numpy.where(ok, cy, 0), safe)
        used = numpy.broadcast_to(nonempty, ix.shape)   # empty tile: no gather
    elif variant == 'stream_r2a':
        ix = numpy.where(ok, cx, 0)
        iy = numpy.where(ok, cy, 0)
        used = numpy.ones_like(ix, dtype=bool)
    elif variant == 'block2':
        first = colsp[k0][:, None]
        ix = numpy.where(e >= lo_e, cx, first)
        iy = numpy.where(e + 1 < hi_e, cy, first)
        used = ok & numpy.ones_like(ix, dtype=bool)
    elif variant == 'pair':
        safe = colsp[k0][:, None]
        ix = numpy.where((e >= lo_e) & (e < hi_e), numpy.where(ok, cx, 0), safe)
        iy = numpy.where(e + 1 < hi_e, numpy.where(ok, cy, 0), safe)
        used = numpy.ones_like(ix, dtype=bool)
    else:
        raise ValueError(variant)
    lo = numpy.minimum(numpy.where(used, ix, big).min(axis=1),
                       numpy.where(used, iy, big).min(axis=1))
    hi = numpy.maximum(numpy.where(used, ix, -1).max(axis=1),
                       numpy.where(used, iy, -1).max(axis=1))
    assert len(lo) == nt
    return lo, hi


def quad_tile_accesses(rowptr, cols, rowblocks):
    '''tile_accesses for the quad-based tiles of the fp16 kernels: per tile
    the smallest / largest index of the gathered vector, and the largest index
    of `cols` (= of the value array) a lane loads.'''
    rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
    cols = numpy.asarray(cols, dtype=numpy.int64)
    rb = numpy.asarray(rowblocks, dtype=numpy.int64)
    quads = (_hip.PMG_NNZ_PER_BLOCK + 4) // (4 * BLOCK)      # kPmgQuads
    k0 = rowptr[rb[:-1]]
    k1 = rowptr[rb[1:]]
    ka = k0 & ~3
    lo_e = (k0 - ka)[:, None]
    hi_e = (k1 - ka)[:, None]
    assert (k1 - ka <= 4 * BLOCK * quads - 1).all(), 'row block larger than tile'
    p = numpy.arange(BLOCK * quads)[None, :]
    loaded = 4 * p < hi_e                       # the quad is loaded at all
    big = numpy.iinfo(numpy.int64).max
    lo = numpy.full(len(k0), big)
    hi = numpy.full(len(k0), -1)
    last_loaded = numpy.where(loaded, ka[:, None] + 4 * p + 3, -1).max(axis=1)
    assert last_loaded.max() < len(cols), \
        'cols / vals must be readable three entries past nnz'
    safe = cols[numpy.minimum(k0, len(cols) - 1)][:, None]
    nonempty = (k0 < k1)[:, None]
    for j in range(4):
        e = 4 * p + j
        idx = numpy.minimum(ka[:, None] + e, len(cols) - 1)
        c = numpy.where(loaded, cols[idx], 0)
        ix = numpy.where((e >= lo_e) & (e < hi_e), c, safe)
        used = numpy.broadcast_to(nonempty, ix.shape)
        lo = numpy.minimum(lo, numpy.where(used, ix, big).min(axis=1))
        hi = numpy.maximum(hi, numpy.where(used, ix, -1).max(axis=1))
    return lo, hi, last_loaded


def cols16_tables(rowptr, cols, rowblocks):
    '''pmg_cols16_kernel (pmg_kernels.hip) in numpy: per row block its lowest
    column (0 for an empty block) and per nonzero the offset from it --
    (cbase, offsets, fits): `fits` = every offset fits in 16 bits.  The kernels
    that read flow_pmg_level.cols16 or the packed streams (flow_mass.packed16,
    flow_pmg_level.packed) rebuild the column as cbase[tile] + offset: with
    these tables that IS the int32 column, so the access sets of
    quad_tile_accesses hold for them unchanged.'''
    rowptr = numpy.asarray(rowptr, dtype=numpy.int64)
    cols = numpy.asarray(cols, dtype=numpy.int64)
    rb = numpy.asarray(rowblocks, dtype=numpy.int64)
    k0, k1 = rowptr[rb[:-1]], rowptr[rb[1:]]
    nnz = int(rowptr[-1])
    tile_of = numpy.repeat(numpy.arange(len(k0)), k1 - k0)
    lo, hi = int(k0[0]), int(k1[-1])
    cbase = numpy.full(len(k0), numpy.iinfo(numpy.int64).max)
    numpy.minimum.at(cbase, tile_of, cols[lo:hi])
    cbase[k0 >= k1] = 0
    off = numpy.zeros(nnz, dtype=numpy.int64)
    off[lo:hi] = cols[lo:hi] - cbase[tile_of]
    return cbase, off, bool((off <= 0xffff).all() and (off >= 0).all())


def window(rowptr, cols, rowblocks, variant='stream'):
    '''[min, max] of x indices over all tiles, or None when nothing is read.'''
    lo, hi = tile_accesses(rowptr, cols, rowblocks, variant)
    sel = lo <= hi
    if not sel.any():
        return None
    return int(lo[sel].min()), int(hi[sel].max())
