"""ctypes loader of libcoslam_hip.so.  Fails loudly when the HIP library is missing: there is no
CPU fallback anywhere in this package (the oracle under oracle/ is test infrastructure only)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COSLAM_HIP_LIB: another build of the same library (A/B runs of two kernel variants on one GPU box)
LIB_PATH = os.environ.get("COSLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libcoslam_hip.so")

_lib = None


class CoslamHipError(RuntimeError):
    pass


def lib():
    """Return the loaded CDLL.  PyTorch (when importable) is imported first so that this library and
    torch share one HIP runtime (same SONAME libamdhip64.so.7) and therefore one set of streams/devices."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CoslamHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  coslam_amd has no CPU fallback."
        )
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:
        pass
    _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    _lib.cs_last_error.restype = ctypes.c_char_p
    _lib.cs_version.restype = ctypes.c_int
    _lib.cs_device_count.restype = ctypes.c_int
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().cs_last_error().decode("utf-8", "replace")
        raise CoslamHipError(f"{what} failed with code {rc}: {msg}")
