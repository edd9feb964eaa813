"""ctypes binding of libnunerf.so (the C-ABI drop-in boundary, include/nu_nerf.h).

The product path has no CPU fallback: if the HIP library is missing or a GPU is absent, the ops
raise.  `import torch` happens first so that the library binds to the HIP runtime torch already
loaded (one runtime => torch's hipStream_t handles are valid inside the library).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# NU_NERF_LIB: another build of the same C ABI (development A/B runs of two kernel generations in one call); default: the in-tree library
_LIB_PATH = os.environ.get("NU_NERF_LIB") or os.path.join(_HERE, "libnunerf.so")
_lib = None

c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_p = ctypes.c_void_p


class NuNerfLibraryError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load libnunerf.so; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NuNerfLibraryError(
                f"{_LIB_PATH} not found: build it with `python -m nu_nerf_amd.build` "
                "(there is no CPU fallback for the product path)")
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return c_p(0)
    return c_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream(device_index=None):
    """The current HIP stream of torch as a void* (what every launch of the library takes).  `torch.cuda.current_stream()` builds
    a Python Stream object per call (~6 us, hundreds of calls per step); the raw getter returns the same handle in ~0.3 us."""
    if _raw_stream is not None:
        return c_p(_raw_stream(torch.cuda.current_device() if device_index is None else device_index))
    return c_p(torch.cuda.current_stream(device_index).cuda_stream)


def check(rc, what):
    if rc != 0:
        raise NuNerfLibraryError(f"{what} failed with code {rc}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NuNerfLibraryError("nu_nerf_amd ops need CUDA(HIP) tensors: there is no CPU fallback")
