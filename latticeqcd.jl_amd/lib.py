"""ctypes binding of liblqcd_hip.so (C ABI: include/lqcd_hip.h).  No CPU fallback: if the shared library is
missing or no HIP device is visible, loading / context creation fails loudly."""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LQCD_HIP_LIB", os.path.join(_HERE, "csrc", "liblqcd_hip.so"))   # override: A/B builds of the same library
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lqcd_hip.h")

OK, ERR_ARG, ERR_HIP, ERR_NOT_CONVERGED, ERR_COMM, ERR_UNSUPPORTED = 0, 1, 2, 3, 4, 5
WILSON, STAGGERED, DOMAINWALL = 0, 1, 2
FULL, EVEN, ODD = 0, 1, 2
LAYOUT_REFERENCE, LAYOUT_DISK = 0, 1

_lib = None


class LQCDError(RuntimeError):
    """Mirrors the reference's `error(...)` (e.g. non-convergent solver, unsupported operator)."""

    def __init__(self, code, msg):
        super().__init__(f"[lqcd_hip status {code}] {msg}")
        self.code = code


class NotConverged(LQCDError):
    pass


def build(verbose=False):
    """Compile liblqcd_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building liblqcd_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))


def declared_symbols():
    """Every function name declared in include/lqcd_hip.h."""
    with open(HEADER) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lqcd_[A-Za-z0-9_]+)\s*\(", src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} not found: run latticeqcd.jl_amd/csrc/build.sh (or __graft_entry__.build()); "
                               "there is no CPU fallback for the hot path")
        L = C.CDLL(SO_PATH)
        L.lqcd_last_error.restype = C.c_char_p
        L.lqcd_index_lex.restype = C.c_int64
        _lib = L
    return _lib


def check(status):
    if status == OK:
        return
    msg = lib().lqcd_last_error().decode("utf-8", "replace")
    if status == ERR_NOT_CONVERGED:
        raise NotConverged(status, msg)
    raise LQCDError(status, msg)


def i4(v):
    return (C.c_int * 4)(*[int(x) for x in v])


def device_count():
    return lib().lqcd_device_count()
