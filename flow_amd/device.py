# -*- coding: utf-8 -*-
'''
Device plumbing: torch is used ONLY for HBM allocation, streams and (in
flow_amd/parallel.py) torch.distributed.  All arithmetic on the path happens
in the HIP library behind the C ABI (include/flow_hip.h); there is no CPU
fallback for it.
'''
import os

import numpy
import torch


def local_rank():
    return int(os.environ.get('LOCAL_RANK', '0'))


_DEVICE = None
_STREAM = None
_POISON = bool(os.environ.get('FLOW_AMD_POISON'))


def get():
    '''cuda:<LOCAL_RANK> when a GPU is visible, else the CPU (host-side setup
    and the CPU-only tests of the host logic).'''
    global _DEVICE
    if _DEVICE is None:
        if torch.cuda.is_available():
            idx = local_rank() % torch.cuda.device_count()
            torch.cuda.set_device(idx)
            _DEVICE = torch.device('cuda', idx)
            # One explicit HIP stream for everything: torch's copies/fills and
            # the kernels launched through the C ABI.  With torch's default
            # stream the handle that crosses the C ABI is 0, which this library
            # takes as the legacy null stream; measured on MI355X (several
            # processes on one GPU) copies issued by torch were then not
            # reliably ordered against those launches.  Same real stream on both
            # sides => plain FIFO order.
            global _STREAM
            _STREAM = torch.cuda.Stream(device=idx)
            torch.cuda.set_stream(_STREAM)
        else:
            _DEVICE = torch.device('cpu')
    return _DEVICE


def on_gpu():
    return get().type == 'cuda'


def to_device(arr):
    arr = numpy.ascontiguousarray(arr)
    if _GUARD and arr.dtype == numpy.float64 and on_gpu():
        t = _guarded(arr.size, None)
        t.copy_(torch.from_numpy(arr.reshape(-1)))
    else:
        t = torch.from_numpy(arr).to(get())
    synchronize()      # uploads are setup; see to_host for why not left async
    return t


_GUARD = 64 if os.environ.get('FLOW_AMD_GUARD') else 0


def _guarded(n, fill):
    '''Debugging aid (FLOW_AMD_GUARD=1): fp64 buffers sit between two 512-byte
    NaN fences, so an out-of-bounds READ that reaches a result shows up.'''
    buf = torch.full((int(n) + 2 * _GUARD,), float('nan'), dtype=torch.float64,
                     device=get())
    view = buf[_GUARD:_GUARD + int(n)]
    if fill is not None:
        view.fill_(fill)
    return view


def zeros(n, dtype=torch.float64):
    if _GUARD and dtype == torch.float64:
        return _guarded(n, 0.0)
    if dtype == torch.float64 and int(n) > 0 and on_gpu():
        # the fill is a kernel of the library on the stream everything else
        # runs on (no torch kernels between the library's: _hip.fill)
        from . import _hip
        return _hip.fill(torch.empty(int(n), dtype=dtype, device=get()), 0.0)
    return torch.zeros(int(n), dtype=dtype, device=get())


def empty(n, dtype=torch.float64):
    if _GUARD and dtype == torch.float64:
        return _guarded(n, float('nan') if _POISON else 0.0)
    if _POISON and dtype == torch.float64:
        # debugging aid: a read of an uninitialised buffer turns into a NaN
        return torch.full((int(n),), float('nan'), dtype=dtype, device=get())
    return torch.empty(int(n), dtype=dtype, device=get())


_HANDLE = None
_FOLLOW_TORCH = bool(os.environ.get('FLOW_AMD_FOLLOW_TORCH_STREAM'))


def stream_handle():
    '''Raw hipStream_t everything is enqueued on (0 on the CPU): the package's
    own stream, which get() also made torch's current one.  The handle is
    cached -- asking torch for its current stream costs ~10 us, and a time step
    asks ~40 times between launches that take ~5 us each.  A caller that
    switches torch's current stream afterwards and wants the library to follow
    sets FLOW_AMD_FOLLOW_TORCH_STREAM=1 (the handle is then looked up every
    time).'''
    global _HANDLE
    if not on_gpu():
        return 0
    if _FOLLOW_TORCH:
        return torch.cuda.current_stream().cuda_stream
    if _HANDLE is None:
        _HANDLE = _STREAM.cuda_stream
    return _HANDLE


def check_stream():
    '''The library enqueues on the package's stream (the cached handle above);
    torch operations of the host code -- masks, uploads, index assignments --
    go to torch's CURRENT stream.  The two are the same stream unless a caller
    has switched torch's (`with torch.cuda.stream(s)`): then nothing orders
    them any more.  Called once per time step: refuse that instead of racing
    (FLOW_AMD_FOLLOW_TORCH_STREAM=1 makes the library follow instead).'''
    if not on_gpu() or _FOLLOW_TORCH or _STREAM is None:
        return
    cur = torch.cuda.current_stream()
    if cur.cuda_stream != _STREAM.cuda_stream:
        raise RuntimeError(
            "flow_amd: torch's current stream (%#x) is not the stream the HIP "
            'library enqueues on (%#x) -- leave the `with torch.cuda.stream(...)` '
            'block, or set FLOW_AMD_FOLLOW_TORCH_STREAM=1 before importing '
            'flow_amd' % (cur.cuda_stream, _STREAM.cuda_stream))


def synchronize():
    if on_gpu():
        torch.cuda.synchronize()


def to_host(t):
    '''Host copy of a device tensor, after a full device synchronisation.
    Kernels are enqueued through the C ABI on torch's current stream; measured
    on MI355X with several processes sharing the GPU, a torch device-to-host
    copy issued right behind them was NOT always ordered after them (stale
    reads in ~40 % of the runs of tests/test_parallel_gpu.py).  Every read-back
    of the package therefore synchronises first.'''
    synchronize()
    return t.detach().cpu()
