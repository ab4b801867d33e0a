"""ctypes front of libetm_envgen.so (csrc/envgen.cc, include/etm_envgen.h): numpy's PCG64 float32 stream restated in C -- the same
floats bit for bit, 4 - 5 x faster than ``Generator.random(dtype=float32)`` -- for the synthetic environment's fresh observation
draws (``pool: 0``).  numpy only: the worker processes of environments/shm_env.py import this module too.

The environment is benchmark infrastructure, not the kernel path: when the library has not been built the environments draw
with numpy itself (identical values, slower) and say so once."""
import ctypes
import os
import sys

import numpy as np

ABI_VERSION = 1
_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_tried = False

# name -> (restype, argtypes); mirrors include/etm_envgen.h one to one
SIGNATURES = {
    "etm_envgen_abi_version": (ctypes.c_int, []),
    "etm_envgen_set_vector": (ctypes.c_int, [ctypes.c_int]),
    "etm_pcg64_fill_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "etm_envgen_pool_create": (ctypes.c_void_p, [ctypes.c_int]),
    "etm_envgen_pool_destroy": (None, [ctypes.c_void_p]),
    "etm_envgen_pool_set_spin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "etm_pcg64_fill_rows_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
}


def path():
    return os.path.join(_HERE, "libetm_envgen.so")


def load(required=False):
    """The library handle, or None when it is not built (``required``: raise instead)."""
    global _lib, _tried
    if _lib is not None or (_tried and not required):
        return _lib
    _tried = True
    try:
        lib = ctypes.CDLL(path())
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.etm_envgen_abi_version() != ABI_VERSION:
            raise OSError(f"libetm_envgen.so ABI {lib.etm_envgen_abi_version()} != expected {ABI_VERSION}; rebuild (make -C csrc)")
        _lib = lib
    except OSError as exc:
        if required:
            raise
        if os.environ.get("ETM_QUIET") != "1":
            print(f"[etm] environments: libetm_envgen.so not usable ({exc}); fresh observations are drawn by numpy (same values, slower)",
                  file=sys.stderr, flush=True)
    return _lib


def state_of(generator):
    """uint64 [4] = (state_hi, state_lo, inc_hi, inc_lo) of a numpy Generator over PCG64 with no buffered 32-bit half."""
    st = generator.bit_generator.state
    if st["bit_generator"] != "PCG64" or st["has_uint32"] != 0:
        raise ValueError("a PCG64 generator without a buffered 32-bit half is needed")
    s, inc, m = st["state"]["state"], st["state"]["inc"], (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m], dtype=np.uint64)


def set_state(generator, state4):
    """The inverse of ``state_of``: numpy's generator continues where the native stream stands."""
    st = generator.bit_generator.state
    st["state"] = {"state": (int(state4[0]) << 64) | int(state4[1]), "inc": (int(state4[2]) << 64) | int(state4[3])}
    st["has_uint32"], st["uinteger"] = 0, 0
    generator.bit_generator.state = st
