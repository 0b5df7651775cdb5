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


def get():
    '''cuda:<LOCAL_RANK> when a GPU is visible, else the CPU (host-side setup
    and the CPU-only tests of the host logic).'''
    global _DEVICE
    if _DEVICE is None:
        if torch.cuda.is_available():
            idx = local_rank() % torch.cuda.device_count()
            torch.cuda.set_device(idx)
            _DEVICE = torch.device('cuda', idx)
        else:
            _DEVICE = torch.device('cpu')
    return _DEVICE


def on_gpu():
    return get().type == 'cuda'


def to_device(arr):
    arr = numpy.ascontiguousarray(arr)
    return torch.from_numpy(arr).to(get())


def zeros(n, dtype=torch.float64):
    return torch.zeros(int(n), dtype=dtype, device=get())


def empty(n, dtype=torch.float64):
    return torch.empty(int(n), dtype=dtype, device=get())


def stream_handle():
    '''Raw hipStream_t of torch's current stream (0 on the CPU).'''
    if on_gpu():
        return torch.cuda.current_stream().cuda_stream
    return 0


def synchronize():
    if on_gpu():
        torch.cuda.synchronize()
