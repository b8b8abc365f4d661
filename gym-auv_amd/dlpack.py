"""Zero-copy hand-off of library-owned device buffers to PyTorch-ROCm.

`auv_dlpack` (C side) allocates the DLManagedTensor record and owns its
deleter, so nothing Python-side is called back when the consumer drops the
tensor (a ctypes callback firing during interpreter teardown would crash).
The tensor aliases the HBM bytes the kernels write: this is how
`obs["perception"]` reaches the LiDAR CNNs (radarCNN.py:275) without a copy.
The buffer lives as long as the simulation handle; keep the `AuvSim` alive
while its tensors are in use.
"""
from __future__ import annotations

import ctypes as C

kDLROCM = 10

_PyCapsule_New = C.pythonapi.PyCapsule_New
_PyCapsule_New.restype = C.py_object
_PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
_NAME = b"dltensor"


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int32),
                ("dtype", DLDataType), ("shape", C.POINTER(C.c_int64)),
                ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class DLManagedTensor(C.Structure):
    """Read-only mirror of the C record, used by tests to inspect an export."""
    _fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p), ("deleter", C.c_void_p)]


def capsule(managed_tensor_ptr: int):
    """Wrap a DLManagedTensor* from `auv_dlpack` in a "dltensor" capsule."""
    if not managed_tensor_ptr:
        raise ValueError("null DLManagedTensor")
    return _PyCapsule_New(managed_tensor_ptr, _NAME, None)


def to_torch(managed_tensor_ptr: int):
    """torch tensor aliasing the exported device buffer (no copy)."""
    import torch.utils.dlpack

    return torch.utils.dlpack.from_dlpack(capsule(managed_tensor_ptr))
