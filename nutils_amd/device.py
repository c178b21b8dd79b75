'''Device plumbing: PyTorch-ROCm is used only as allocator / stream owner /
``torch.distributed`` (RCCL) front-end.  All compute goes through the C ABI of
libnutils_hip.so with raw device pointers.'''

import ctypes
import numpy

from . import _lib


def torch():
    import torch as _torch
    return _torch


def require_gpu():
    t = torch()
    if not t.cuda.is_available():
        raise _lib.NutilsHipError('no HIP device visible: nutils_amd runs its element loop on an MI355X and has no CPU fallback')
    _lib.load()
    return t


def current_device():
    return torch().cuda.current_device()


def stream():
    '''hipStream_t of torch's current stream, as an integer for ctypes.'''
    return ctypes.c_void_p(torch().cuda.current_stream().cuda_stream)


_NP2T = {'float64': 'float64', 'int32': 'int32', 'int64': 'int64', 'uint8': 'uint8'}


def to_dev(array, dtype):
    '''Copy a host array to HBM as a contiguous tensor of `dtype` (numpy dtype name).'''
    t = require_gpu()
    src = numpy.asarray(array)
    if src.size * numpy.dtype(dtype).itemsize >= 1 << 20:
        # large arrays (field coefficients of a Newton step, vertex arrays): one conversion straight into page-locked memory from torch's
        # caching host allocator and an asynchronous copy (the allocator keeps the block until the copy has run); a pageable copy runs at ~8 GB/s
        host = t.empty(src.shape, dtype=getattr(t, _NP2T[dtype]), pin_memory=True)
        numpy.copyto(host.numpy(), src, casting='unsafe')
        return host.to(device='cuda', non_blocking=True)
    a = numpy.array(array, dtype=dtype, order='C', copy=True)
    return t.from_numpy(a).to(device='cuda', non_blocking=False)


def empty(n, dtype):
    t = require_gpu()
    return t.empty(int(n), dtype=getattr(t, _NP2T[dtype]), device='cuda')


def zeros(n, dtype):
    t = require_gpu()
    return t.zeros(int(n), dtype=getattr(t, _NP2T[dtype]), device='cuda')


def ptr(tensor):
    '''Raw device pointer (or NULL) for the C ABI.'''
    if tensor is None:
        return None
    return ctypes.c_void_p(tensor.data_ptr())


def host_ptr(array):
    if array is None:
        return None
    return ctypes.c_void_p(array.ctypes.data)


def to_host(tensor):
    '''Fresh host (numpy) copy.  Large arrays go through page-locked memory from torch's caching host allocator (a pageable
    `.cpu()` copy of the 0.2 GB C4 Jacobian runs at ~8 GB/s, the pinned one at PCIe speed); the block returns to the cache when
    the numpy array is released.'''
    if tensor.is_cuda and tensor.numel() * tensor.element_size() >= 1 << 20:
        host = torch().empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
        host.copy_(tensor)
        return host.numpy()
    return tensor.cpu().numpy()


def synchronize():
    torch().cuda.synchronize()
