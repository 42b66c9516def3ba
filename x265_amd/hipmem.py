"""Device buffers for the `_device` entry points without PyTorch: hipMalloc / hipMemcpy / hipFree through the HIP
runtime that libx265_hip.so itself is linked to (a second copy of the runtime in one process, e.g. the one
PyTorch bundles, does not see the GPU once the first has initialised it).  Plumbing for tests and the sequence
runner; nothing here is on a timed path."""
import ctypes as C

import numpy as np

from . import lib

H2D, D2H = 1, 2


class HipError(RuntimeError):
    pass


def _rt():
    h = lib.load()
    h.hipGetErrorString.restype = C.c_char_p
    return h


def _check(rc, what):
    if rc != 0:
        raise HipError(f"{what}: {_rt().hipGetErrorString(C.c_int(rc)).decode()}")


class DeviceBuffer:
    """`nbytes` of device memory on the current device; `ptr` is the raw device address."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(_rt().hipMalloc(C.byref(p), C.c_size_t(max(int(nbytes), 1))), "hipMalloc")
        self.ptr, self.nbytes = p.value, int(nbytes)

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.upload(a)
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        _check(_rt().hipMemcpy(C.c_void_p(self.ptr), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.c_int(H2D)), "hipMemcpy H2D")

    def download(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(count, dtype)
        _check(_rt().hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), C.c_int(D2H)), "hipMemcpy D2H")
        return out

    def fill(self, byte):
        _check(_rt().hipMemset(C.c_void_p(self.ptr), C.c_int(byte), C.c_size_t(self.nbytes)), "hipMemset")

    def free(self):
        if self.ptr:
            _rt().hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
