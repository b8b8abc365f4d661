"""Zero-copy export of library-owned device buffers as DLPack capsules.

The capsule borrows the memory: its deleter only drops the Python-side
bookkeeping, the buffer itself lives as long as the simulation handle.  The
consumer (PyTorch-ROCm) sees the same HBM bytes the kernels write, which is
how `obs["perception"]` reaches the LiDAR CNNs (radarCNN.py:275) without a
copy.
"""
from __future__ import annotations

import ctypes as C

from . import _lib

kDLROCM = 10
_CODE = {_lib.F64: (2, 64), _lib.F32: (2, 32), _lib.I32: (0, 32), _lib.U8: (1, 8), _lib.I64: (0, 64)}


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int32),
                ("dtype", DLDataType), ("shape", C.POINTER(C.c_int64)),
                ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p),
                            ("deleter", _DELETER)]

_alive = {}  # address of DLManagedTensor -> objects that must outlive the consumer


@_DELETER
def _deleter(ptr):
    _alive.pop(C.addressof(ptr.contents), None)


_PyCapsule_New = C.pythonapi.PyCapsule_New
_PyCapsule_New.restype = C.py_object
_PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
_NAME = b"dltensor"


def to_capsule(dev_ptr: int, shape, dtype: int, device_id: int, owner=None):
    """DLPack capsule for a dense row-major device buffer."""
    code, bits = _CODE[dtype]
    nd = len(shape)
    shape_arr = (C.c_int64 * nd)(*[int(s) for s in shape])
    mt = DLManagedTensor()
    mt.dl_tensor.data = dev_ptr
    mt.dl_tensor.device = DLDevice(kDLROCM, device_id)
    mt.dl_tensor.ndim = nd
    mt.dl_tensor.dtype = DLDataType(code, bits, 1)
    mt.dl_tensor.shape = C.cast(shape_arr, C.POINTER(C.c_int64))
    mt.dl_tensor.strides = None
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _deleter
    _alive[C.addressof(mt)] = (mt, shape_arr, owner)
    return _PyCapsule_New(C.addressof(mt), _NAME, None)


def to_torch(dev_ptr: int, shape, dtype: int, device_id: int, owner=None):
    """torch tensor aliasing the device buffer (no copy)."""
    import torch.utils.dlpack

    return torch.utils.dlpack.from_dlpack(to_capsule(dev_ptr, shape, dtype, device_id, owner))
