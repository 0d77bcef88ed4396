"""ctypes binding of libsed_hip.so, generated from include/sed_hip.h (the single source of truth).

The product path has NO fallback: if the library is missing or fails to load, `lib()` raises.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "sed_hip.h")
TEST_HEADER = os.path.join(os.path.dirname(HERE), "include", "sed_hip_test.h")       # test hooks: not the product ABI
LIB_PATH = os.path.join(HERE, "libsed_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "sed_stream_t": ctypes.c_void_p, "const char*": ctypes.c_char_p,
}


def _ctype_of(decl):
    t = decl.strip()
    t = re.sub(r"\s+\w+$", "", t) if not t.endswith("*") else t          # drop the parameter name
    t = re.sub(r"\s*\*\s*", "*", t)
    if t.endswith("*"):
        return ctypes.c_char_p if t == "const char*" else ctypes.c_void_p
    return _CTYPES[t]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every `sed_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int|long)\s+(sed_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?[\*\s])(\w+)$", a)
                argtypes.append(_ctype_of(mm.group(1)))
        protos[name] = (_CTYPES.get(ret, ctypes.c_char_p) if ret != "const char*" else ctypes.c_char_p, argtypes)
    return protos


_LIB = None


def _load_hip_runtime():
    """libsed_hip.so carries no DT_NEEDED for the HIP runtime (see build.py): make the process-wide runtime --
    PyTorch-ROCm's bundled libamdhip64.so -- globally visible before dlopen()ing it."""
    import torch  # noqa: F401  (loads its HIP runtime)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    for path in (cand, "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            return
        except OSError:
            continue
    raise RuntimeError("no HIP runtime (libamdhip64.so) could be loaded")


def lib():
    """Load (once) and return the ctypes handle with prototypes installed.  Raises if unavailable."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsed_hip.so not found at %s — build it with `python -m sound_event_detection_dcase2017_task4_amd.build` "
                "(hipcc, gfx950).  There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
        _load_hip_runtime()
        h = ctypes.CDLL(LIB_PATH)
        for name, (ret, argtypes) in parse_header().items():
            fn = getattr(h, name)            # AttributeError if the library lacks a declared symbol
            fn.restype = ret
            fn.argtypes = argtypes
        verify_flags(h)
        _LIB = h
    return _LIB


def test_hooks():
    """The library handle with the prototypes of include/sed_hip_test.h installed as well (tests and tools only)."""
    h = lib()
    for name, (ret, argtypes) in parse_header(TEST_HEADER).items():
        fn = getattr(h, name)
        fn.restype = ret
        fn.argtypes = argtypes
    return h


def verify_flags(h):
    """Refuse a library that was not compiled with the default flags of build.py (sed_version() carries the hash of the flags
    of every object): a timing-experiment build (-DSF_ABL_..., tools/experiments/*.patch) computes wrong results by design
    and must never be picked up by accident.  SED_ALLOW_EXPERIMENT=1 lets the experiment tooling load it."""
    from . import build
    fn = h.sed_version
    fn.restype = ctypes.c_char_p
    ver = fn().decode()
    want = build.flags_hash(build.BASE_FLAGS)
    got = ver.split("flags:")[-1] if "flags:" in ver else "none"
    if got != want and os.environ.get("SED_ALLOW_EXPERIMENT") != "1":
        raise RuntimeError(
            "libsed_hip.so was built with non-default hipcc flags (%r, default build = %s): an experiment build is refused.  "
            "Rebuild with `python -m sound_event_detection_dcase2017_task4_amd.build --force` (SED_HIPCC_FLAGS unset), or set "
            "SED_ALLOW_EXPERIMENT=1 to load it on purpose." % (ver, want))
    return ver


class SedHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise SedHipError("%s failed with code %d%s" % (what, rc, " (invalid argument)" if rc == -22 else " (hipError_t)"))
