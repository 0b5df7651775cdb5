# -*- coding: utf-8 -*-
'''
ctypes binding of libflow_hip.so (C ABI: include/flow_hip.h).

There is NO fallback: if the library is missing, or a call is made without a
GPU, this raises.  Wrappers check operand dtypes / sizes / devices on the host
before a kernel is launched (an out-of-bounds access on the device can take
the whole node down).
'''
import ctypes
import os

import torch

from . import device

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libflow_hip.so')

REDUCE_WORK = 4096
GMRES_MAX_RESTART = 30
GMRES_PARTIALS = (GMRES_MAX_RESTART + 2) * 1024
GMRES_STATE = 1280
SPMV_ROWS_PER_BLOCK = 256
SPMV_NNZ_PER_BLOCK = 1022
PMG_NNZ_PER_BLOCK = 2044

c_double_p = ctypes.c_void_p
c_int_p = ctypes.c_void_p


class Operator(ctypes.Structure):
    _fields_ = [
        ('kind', ctypes.c_int),
        ('n', ctypes.c_int),
        ('nnz', ctypes.c_int),
        ('nblocks', ctypes.c_int),
        ('rowptr', ctypes.c_void_p),
        ('cols', ctypes.c_void_p),
        ('rowblocks', ctypes.c_void_p),
        ('vals', ctypes.c_void_p * 4),
        ('matfree', ctypes.c_void_p),
        ('rowmask', ctypes.c_void_p),
        ]


class CoarseS(ctypes.Structure):
    _fields_ = [
        ('n', ctypes.c_int),
        ('nc', ctypes.c_int),
        ('agg_ptr', ctypes.c_void_p),
        ('agg_dofs', ctypes.c_void_p),
        ('agg_of', ctypes.c_void_p),
        ('lda', ctypes.c_int),
        ('Ainv', ctypes.c_void_p),
        ]


MG_MAX_LEVELS = 8


class MgS(ctypes.Structure):
    _fields_ = [
        ('nlevels', ctypes.c_int),
        ('Ah', Operator * MG_MAX_LEVELS),
        ('dinv', ctypes.c_void_p * MG_MAX_LEVELS),
        ('Ps', Operator * MG_MAX_LEVELS),
        ('R', Operator * MG_MAX_LEVELS),
        ('r', ctypes.c_void_p * MG_MAX_LEVELS),
        ('x', ctypes.c_void_p * MG_MAX_LEVELS),
        ('t', ctypes.c_void_p * MG_MAX_LEVELS),
        ('nc', ctypes.c_int), ('lda', ctypes.c_int),
        ('Ainv', ctypes.c_void_p),
        ('omega', ctypes.c_double),
        ('C', Operator * MG_MAX_LEVELS),
        ('up_rowblocks', ctypes.c_void_p * MG_MAX_LEVELS),
        ('up_nblocks', ctypes.c_int * MG_MAX_LEVELS),
        ]


class IluPlanS(ctypes.Structure):
    _fields_ = [
        ('n', ctypes.c_int), ('nnz', ctypes.c_int), ('ncolors', ctypes.c_int),
        ('nnz_l', ctypes.c_int), ('nnz_u', ctypes.c_int),
        ('off_l', ctypes.c_int), ('off_u', ctypes.c_int),
        ('off_d', ctypes.c_int), ('lu_size', ctypes.c_int),
        ('max_row', ctypes.c_int), ('nslices', ctypes.c_int),
        ('color_ptr_host', ctypes.c_void_p),
        ('slice_ptr_host', ctypes.c_void_p),
        ('rowptr', ctypes.c_void_p), ('cols', ctypes.c_void_p),
        ('diag', ctypes.c_void_p), ('src_pos', ctypes.c_void_p),
        ('old_of_new', ctypes.c_void_p), ('new_of_old', ctypes.c_void_p),
        ('slice_row', ctypes.c_void_p),
        ('l_slice_off', ctypes.c_void_p), ('l_cols', ctypes.c_void_p),
        ('l_pos', ctypes.c_void_p),
        ('u_slice_off', ctypes.c_void_p), ('u_cols', ctypes.c_void_p),
        ('u_pos', ctypes.c_void_p),
        ]


class IluS(ctypes.Structure):
    _fields_ = [
        ('plan', ctypes.POINTER(IluPlanS)),
        ('nblocks', ctypes.c_int),
        ('lu', ctypes.c_void_p),
        ('packed', ctypes.c_void_p),
        ('single_vector', ctypes.c_int),
        ('cycle', ctypes.c_void_p),
        ]


class TlS(ctypes.Structure):
    '''flow_tl (include/flow_hip.h, K19)'''
    _fields_ = [
        ('fine', ctypes.POINTER(IluS)), ('coarse', ctypes.POINTER(IluS)),
        ('fine_op', ctypes.c_void_p), ('coarse_op', ctypes.c_void_p),
        ('pre', ctypes.c_int), ('post', ctypes.c_int),
        ('coarse_sweeps', ctypes.c_int),
        ('ends', ctypes.c_void_p), ('rptr', ctypes.c_void_p),
        ('rsrc', ctypes.c_void_p),
        ('bc_fine', ctypes.c_void_p), ('bc_coarse', ctypes.c_void_p),
        ('rscale', ctypes.c_void_p), ('work', ctypes.c_void_p),
        ]


class PmgLevelS(ctypes.Structure):
    _fields_ = [
        ('n', ctypes.c_int), ('nnz', ctypes.c_int), ('nblocks', ctypes.c_int),
        ('rowptr', ctypes.c_void_p), ('cols', ctypes.c_void_p),
        ('rowblocks', ctypes.c_void_p),
        ('vals', ctypes.c_void_p), ('diag', ctypes.c_void_p),
        ('dinv', ctypes.c_void_p),
        ('lam_min', ctypes.c_double), ('lam_max', ctypes.c_double),
        ('cols16', ctypes.c_void_p), ('cbase', ctypes.c_void_p),
        ('packed', ctypes.c_void_p), ('idrows', ctypes.c_void_p),
        ]


class PmgS(ctypes.Structure):
    _fields_ = [
        ('fine', PmgLevelS), ('coarse', PmgLevelS),
        ('pre', ctypes.c_int), ('post', ctypes.c_int),
        ('coarse_steps', ctypes.c_int),
        ('ends', ctypes.c_void_p),
        ('rptr', ctypes.c_void_p), ('rsrc', ctypes.c_void_p),
        ('bc_fine', ctypes.c_void_p), ('bc_coarse', ctypes.c_void_p),
        ('work', ctypes.c_void_p),
        ('scalar', ctypes.c_int),
        ]


class MassS(ctypes.Structure):
    _fields_ = [
        ('A', ctypes.POINTER(Operator)),
        ('dinv', ctypes.c_void_p),
        ('nblocks16', ctypes.c_int),
        ('rowblocks16', ctypes.c_void_p),
        ('vals16', ctypes.c_void_p),
        ('lam_min', ctypes.c_double), ('lam_max', ctypes.c_double),
        ('steps', ctypes.c_int),
        ('contraction', ctypes.c_double),
        ('work16', ctypes.c_void_p),
        ('packed16', ctypes.c_void_p),
        ('cbase16', ctypes.c_void_p),
        ('work16_rows', ctypes.c_int),
        ]


class MassStripsS(ctypes.Structure):
    _fields_ = [
        ('nlevels', ctypes.c_int),
        ('rowblocks16', ctypes.c_void_p * 16),
        ('nblocks16', ctypes.c_int * 16),
        ('row_lo_last', ctypes.c_int), ('row_hi_last', ctypes.c_int),
        ]


class MeshS(ctypes.Structure):
    _fields_ = [('nc', ctypes.c_int), ('xy', ctypes.c_void_p),
                ('c0', ctypes.c_int), ('c1', ctypes.c_int)]


class SpaceS(ctypes.Structure):
    _fields_ = [
        ('deg', ctypes.c_int),
        ('n', ctypes.c_int),
        ('nnz', ctypes.c_int),
        ('cell_dofs', ctypes.c_void_p),
        ('cptr', ctypes.c_void_p),
        ('csrc', ctypes.c_void_p),
        ('vptr', ctypes.c_void_p),
        ('vsrc', ctypes.c_void_p),
        ('r0', ctypes.c_int), ('r1', ctypes.c_int),
        ('nnz0', ctypes.c_int), ('nnz1', ctypes.c_int),
        ]


class CoefS(ctypes.Structure):
    _fields_ = [
        ('nl', ctypes.c_int),
        ('cell_stride', ctypes.c_int),
        ('values', ctypes.c_void_p),
        ('G', ctypes.c_void_p),
        ]


class NsParams(ctypes.Structure):
    _fields_ = [
        ('dt', ctypes.c_double), ('rho', ctypes.c_double),
        ('mu', ctypes.c_double), ('theta_i', ctypes.c_double),
        ('theta_e', ctypes.c_double),
        ]


ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)


class PeerS(ctypes.Structure):
    '''flow_peer (include/flow_hip.h): halos from neighbour to neighbour'''
    _fields_ = [
        ('flags', ctypes.c_void_p), ('land', ctypes.c_void_p),
        ('land_cap', ctypes.c_int), ('spin_limit', ctypes.c_int),
        ('nb_flags', ctypes.c_void_p * 2), ('nb_land', ctypes.c_void_p * 2),
        ('seq_host', ctypes.POINTER(ctypes.c_ulonglong * 5)),
        ]


PEER_FLAGS = 16


class CommS(ctypes.Structure):
    _fields_ = [
        ('rank', ctypes.c_int), ('world', ctypes.c_int),
        ('buf', ctypes.c_void_p), ('capacity', ctypes.c_int),
        ('allreduce', ALLREDUCE_FN), ('user', ctypes.c_void_p),
        ('peer', ctypes.POINTER(PeerS)),
        ]


class RcclBinding(ctypes.Structure):
    _fields_ = [('comm', ctypes.c_void_p), ('buf', ctypes.c_void_p),
                ('stream', ctypes.c_void_p)]


class RowsS(ctypes.Structure):
    _fields_ = [
        ('n', ctypes.c_int), ('r0', ctypes.c_int), ('r1', ctypes.c_int),
        ('e0', ctypes.c_int), ('e1', ctypes.c_int), ('nhalo', ctypes.c_int),
        ('send_row', ctypes.c_int * 2), ('send_len', ctypes.c_int * 2),
        ('send_slot', ctypes.c_int * 2),
        ('recv_row', ctypes.c_int * 2), ('recv_len', ctypes.c_int * 2),
        ('recv_slot', ctypes.c_int * 2),
        ]


class MgShardS(ctypes.Structure):
    _fields_ = [
        ('mg', ctypes.POINTER(MgS)),
        ('Ah0', Operator), ('Ps0', Operator), ('Rg', Operator),
        ('Cg', Operator),
        ('up_rowblocks0', ctypes.c_void_p), ('up_nblocks0', ctypes.c_int),
        ('z_lo', ctypes.c_int), ('z_hi', ctypes.c_int),
        ]


class MomentumJvp(ctypes.Structure):
    _fields_ = [
        ('mesh', ctypes.POINTER(MeshS)),
        ('W', ctypes.POINTER(SpaceS)),
        ('bfmask', ctypes.c_void_p),
        ('ui', ctypes.c_void_p),
        ('prm', NsParams),
        ('scratch', ctypes.c_void_p),
        ('nbc', ctypes.c_int),
        ('bc_dofs', ctypes.c_void_p),
        ('bc_mask', ctypes.c_void_p),
        ]


# every symbol include/flow_hip.h declares: (name, argtypes)
_VP = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
_P = ctypes.POINTER
SYMBOLS = {
    'flow_abi_version': [],
    'flow_launch_count': [_P(ctypes.c_ulonglong)],
    'flow_graph_mode': [_I, ctypes.c_longlong],
    'flow_graph_stats': [_P(ctypes.c_ulonglong)],
    'flow_xcd_tile_host': [_I, _I],
    'flow_spmv_tile_nnz': [_I],
    'flow_operator_apply': [_P(Operator), _VP, _VP, _VP],
    'flow_profile_spmv_begin': [_I, _I],
    'flow_profile_spmv_end': [_P(_D), _P(_I)],
    'flow_profile_event_overhead': [_P(_D), _VP],
    'flow_profile_marker': [_I, _VP],
    'flow_profile_stream_copy': [ctypes.c_size_t, _VP, _VP, _VP],
    'flow_profile_stream_read': [ctypes.c_size_t, _VP, _VP, _VP],
    'flow_operator_diag_inv': [_P(Operator), _VP, _VP, _VP],
    'flow_dot_host': [_I, _VP, _VP, _VP, _P(_D), _VP],
    'flow_norm_host': [_I, _VP, _I, _VP, _P(_D), _VP],
    'flow_axpby': [_I, _D, _VP, _D, _VP, _VP],
    'flow_vmul': [_I, _D, _VP, _VP, _VP, _VP],
    'flow_fill': [_I, _D, _VP, _VP],
    'flow_lincomb': [_I, _I, _P(_D), _P(_VP), _VP, _VP],
    'flow_fingerprint': [_I, _P(_VP), _P(_I), _VP, _VP, _VP],
    'flow_read_doubles': [_VP, _I, _P(_D), _VP],
    'flow_extrapolation_weights': [_I, _P(_D), _D, _I, _I, _P(_D)],
    'flow_scale_rows': [_I, _VP, _VP, _VP, _VP],
    'flow_cg_solve': [_P(Operator), _VP, _P(CoarseS), _P(MgS), _VP, _VP, _D, _D,
                      _I, _I, _I, _VP, ctypes.c_size_t, _P(_I), _P(_D), _VP],
    'flow_cg_solve_guarded': [_P(Operator), _VP, _P(CoarseS), _P(MgS), _VP, _VP,
                              _VP, _D, _D, _I, _I, _I, _VP, ctypes.c_size_t,
                              _P(_I), _P(_D), _P(_I), _VP],
    'flow_mg_apply': [_P(MgS), _I, _VP, _VP, _VP],
    'flow_two_level_apply': [_P(CoarseS), _VP, _VP, _VP, _VP, _VP],
    'flow_bicgstab_solve': [_P(Operator), _VP, _P(IluS), _VP, _VP, _D, _D, _I,
                            _I, _I, _VP, ctypes.c_size_t, _P(_I), _P(_D), _VP],
    'flow_gmres_solve': [_P(Operator), _VP, _P(IluS), _P(PmgS), _VP, _VP, _D,
                         _D, _I, _I, _I, _I, _I, _VP, ctypes.c_size_t, _P(_I),
                         _P(_D), _VP],
    'flow_pmg_pack': [_I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    'flow_pmg_pack1': [_I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                       _VP, _VP, _VP, _VP],
    'flow_pmg_cols16': [_I, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    'flow_pmg_lambda_max': [_P(PmgLevelS), _I, _VP, _VP, _VP, _P(_D), _VP],
    'flow_pmg_apply': [_P(PmgS), _VP, _VP, _VP],
    'flow_tl_apply': [_P(TlS), _VP, _VP, _VP],
    'flow_aggregate_host': [_I, _VP, _VP, _VP, ctypes.c_double, _VP, _VP,
                            _P(ctypes.c_int)],
    'flow_peer_alloc': [ctypes.c_int, _P(ctypes.c_void_p), ctypes.c_char_p],
    'flow_peer_open': [ctypes.c_char_p, _P(ctypes.c_void_p)],
    'flow_peer_close': [_VP],
    'flow_peer_free': [_VP],
    'flow_peer_status': [_P(PeerS), _P(ctypes.c_ulonglong), _VP],
    'flow_mass_pack': [_I, _VP, _VP, _VP, _VP, _VP],
    'flow_mass_pack16': [_I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    'flow_mass_solve': [_P(MassS), _VP, _VP, _D, _D, _I, _I, _VP,
                        ctypes.c_size_t, _P(_I), _P(_D), _VP],
    'flow_mass_solve_increment': [_P(MassS), _VP, _VP, _VP, _VP, _D, _D, _I, _I, _VP,
                                  ctypes.c_size_t, _P(_I), _P(_D), _VP],
    'flow_gather_rows': [_I, _VP, _I, _VP, _I, _VP, _I, _VP],
    'flow_color_greedy_host': [_I, _VP, _VP, _VP, _P(_I)],
    'flow_color_iterate_host': [_I, _VP, _VP, _VP, _P(_I), _I],
    'flow_ilu0_factor': [_P(IluPlanS), _I, _VP, _VP, _VP, _VP],
    'flow_ilu0_pack': [_P(IluS), _VP, _VP],
    'flow_ilu0_solve': [_P(IluS), _VP, _VP, _VP, _VP],
    'flow_rccl_load': [ctypes.c_char_p],
    'flow_rccl_unique_id': [_VP],
    'flow_rccl_comm_create': [_VP, _I, _I, _P(_VP)],
    'flow_rccl_comm_destroy': [_VP],
    'flow_rccl_allreduce': [_VP, _I],
    'flow_shard_halo': [_P(CommS), _P(RowsS), _I, _VP, _I, _VP],
    'flow_shard_reduce_host': [_P(CommS), _P(RowsS), _I, _VP, _VP, _I, _I, _VP,
                               _P(_D), _VP],
    'flow_shard_cg_solve': [_P(CommS), _P(RowsS), _P(Operator), _VP, _VP, _VP,
                            _D, _D, _I, _I, _I, _VP, ctypes.c_size_t, _P(_I),
                            _P(_D), _P(_I), _VP],
    'flow_shard_mgcg_solve': [_P(CommS), _P(RowsS), _P(Operator), _VP,
                              _P(MgShardS), _VP, _VP, _D, _D, _I, _I, _I, _VP,
                              ctypes.c_size_t, _P(_I), _P(_D), _P(_I), _VP],
    'flow_shard_mass_solve': [_P(CommS), _P(RowsS), _P(MassS), _P(MassStripsS),
                              _VP, _VP, _VP, _VP, _D, _D, _I, _I, _VP,
                              ctypes.c_size_t, _P(_I), _P(_D), _VP],
    'flow_shard_gmres_solve': [_P(CommS), _P(RowsS), _P(Operator), _P(IluS),
                               _P(PmgS), _VP, _VP, _D, _D, _I, _I, _I, _I, _I,
                               _VP, ctypes.c_size_t, _P(_I), _P(_D), _VP],
    'flow_assemble_scalar_matrix': [_I, _P(MeshS), _P(SpaceS), _VP, _VP, _VP],
    'flow_assemble_pressure_rhs': [_P(MeshS), _P(SpaceS), _P(SpaceS), _VP, _VP,
                                   _D, _D, _I, _VP, _VP, _VP],
    'flow_assemble_correction_rhs': [_P(MeshS), _P(SpaceS), _P(SpaceS), _VP,
                                     _VP, _VP, _D, _D, _I, _VP, _VP, _VP],
    'flow_assemble_momentum': [_P(MeshS), _P(SpaceS), _P(SpaceS), _VP, _VP, _VP,
                               _VP, _P(CoefS), _P(CoefS), _P(NsParams), _VP,
                               _VP, _VP, ctypes.c_size_t, _VP],
    'flow_momentum_jvp_apply': [_P(MomentumJvp), _VP, _VP, _VP],
    'flow_assemble_source': [_P(MeshS), _P(SpaceS), _I, _P(CoefS), _VP, _VP,
                             _VP],
    'flow_assemble_magnitude': [_P(MeshS), _P(SpaceS), _I, _VP, _VP, _VP, _VP],
    'flow_assemble_div_adjoint': [_P(MeshS), _P(SpaceS), _P(SpaceS), _VP, _VP,
                                  _VP, _VP],
    'flow_bc_identity_rows': [_P(Operator), _VP, _VP, _I, _VP, _VP],
    'flow_bc_residual': [_I, _VP, _VP, _VP, _VP, _VP],
    'flow_bc_set_values': [_I, _VP, _VP, _VP, _VP],
    'flow_bc_symmetric_matrix': [_I, _VP, _VP, _VP, _VP, _VP, _VP],
    'flow_assemble_heat': [_P(MeshS), _P(SpaceS), _P(SpaceS), _VP, _D, _D, _I,
                           _VP, _VP, _VP, _VP, _VP, _VP],
    'flow_assemble_heat_supg_source': [_P(MeshS), _P(SpaceS), _P(SpaceS), _VP,
                                       _D, _D, _P(CoefS), _VP, _VP, _VP, _VP],
    }

_LIB = None


class HipError(RuntimeError):
    pass


class NotConverged(RuntimeError):
    '''Krylov / Newton non-convergence.  A RuntimeError, because callers of the
    reference catch exactly that (tests/test_boussinesq.py:254).'''


# flow_abi_version() of the library these bindings describe (the structs above
# and SYMBOLS): a stale libflow_hip.so is refused at load time
ABI_VERSION = 30


def load_library():
    '''dlopen the in-tree library and set the prototypes (no GPU needed).'''
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise HipError(
                'libflow_hip.so not found at %s -- build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C flow_amd/csrc`' % LIB_PATH
                )
        lib = ctypes.CDLL(LIB_PATH)
        lib.flow_last_error.restype = ctypes.c_char_p
        lib.flow_last_error.argtypes = []
        for name, argtypes in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = argtypes
        got = lib.flow_abi_version()
        if got != ABI_VERSION:
            raise HipError(
                '%s is ABI version %d, these bindings are version %d: '
                'rebuild it (`make -C flow_amd/csrc`)'
                % (LIB_PATH, got, ABI_VERSION))
        _LIB = lib
    return _LIB


def launch_count():
    '''Kernel launches the library has issued so far (no GPU needed to ask).'''
    n = ctypes.c_ulonglong(0)
    check(load_library().flow_launch_count(ctypes.byref(n)))
    return int(n.value)


_GRAPHS = {'mode': None}


def graph_mode(mode, auto_rows=-1, sites=0):
    '''Iteration bodies as HIP graphs: 0 never, 1 always, 2 by system size;
    sites: 1 CG | 2 GMRES | 4 mass solver (0: leave) (include/flow_hip.h:
    flow_graph_mode).'''
    check(load_library().flow_graph_mode(int(mode) | (int(sites) << 4),
                                         int(auto_rows)))
    _GRAPHS['mode'] = int(mode)


def graphs_possible():
    '''Can a solver loop be replayed as a HIP graph in this process (the
    option is off unless FLOW_AMD_GRAPHS or graph_mode() says otherwise)?
    Host code that keeps operands at fixed addresses only for the replay's
    sake asks here first.'''
    if _GRAPHS['mode'] is not None:
        return _GRAPHS['mode'] != 0
    return os.environ.get('FLOW_AMD_GRAPHS', '0') not in ('', '0')


def graph_stats():
    '''dict: graphs kept, captures, replays, kernel nodes replayed.'''
    v = (ctypes.c_ulonglong * 8)()
    check(load_library().flow_graph_stats(v))
    return {'graphs': int(v[0]), 'captures': int(v[1]), 'replays': int(v[2]),
            'nodes': int(v[3]), 'captures_cg': int(v[4]),
            'captures_gmres': int(v[5]), 'captures_mass': int(v[6])}


def spmv_tile_nnz(kind=0):
    '''Nonzeros a CSR-stream row block of an operator of `kind` may hold (a
    build constant of the library; no GPU needed).'''
    return int(load_library().flow_spmv_tile_nnz(int(kind)))


def lib():
    '''The library, for compute calls: requires a GPU.'''
    handle = load_library()
    if not device.on_gpu():
        raise HipError(
            'flow_amd needs an AMD GPU (MI355X): the hot path has no CPU '
            'fallback'
            )
    return handle


def check(rc):
    if rc == 0:
        return
    msg = load_library().flow_last_error().decode('utf-8', 'replace')
    if rc == 1:
        raise NotConverged(msg)
    if rc == 2:
        raise ValueError(msg)
    raise HipError(msg)


# -- operand checks -----------------------------------------------------------
def _ptr(t, dtype, numel=None, name='operand'):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s: expected a torch tensor' % name)
    if t.dtype != dtype:
        raise TypeError('%s: dtype %s, expected %s' % (name, t.dtype, dtype))
    if not t.is_cuda:
        raise HipError('%s is not in device memory' % name)
    if not t.is_contiguous():
        raise ValueError('%s is not contiguous' % name)
    if numel is not None and t.numel() < numel:
        raise ValueError(
            '%s has %d elements, needs %d' % (name, t.numel(), numel)
            )
    return ctypes.c_void_p(t.data_ptr())


def f64(t, numel=None, name='fp64 operand'):
    return _ptr(t, torch.float64, numel, name)


def f32(t, numel=None, name='fp32 operand'):
    return _ptr(t, torch.float32, numel, name)


def f16(t, numel=None, name='fp16 operand'):
    return _ptr(t, torch.float16, numel, name)


def i32(t, numel=None, name='int32 operand'):
    return _ptr(t, torch.int32, numel, name)


def u8(t, numel=None, name='uint8 operand'):
    return _ptr(t, torch.uint8, numel, name)


def stream():
    return ctypes.c_void_p(device.stream_handle())


# -- moving data on the device ------------------------------------------------
# Fills and device-to-device copies of fp64 vectors go through kernels of the
# library on the stream everything else runs on (flow_fill / flow_axpby), not
# through the runtime's memset / memcpy: see fill_kernel in la_kernels.hip.
# Host tensors (CPU-only setup and tests) take torch's path.
def copy(dst, src):
    assert dst.numel() == src.numel()
    if not dst.is_cuda:
        dst.copy_(src)
    elif dst.data_ptr() != src.data_ptr():
        check(lib().flow_axpby(dst.numel(), 1.0, f64(src, name='src'), 0.0,
                               f64(dst, name='dst'), stream()))
    return dst


def fill(t, value=0.0):
    if not t.is_cuda:
        t.fill_(float(value))
    elif t.numel() > 0:
        check(lib().flow_fill(t.numel(), float(value), f64(t), stream()))
    return t


def clone(t):
    out = torch.empty_like(t)
    return copy(out, t)
