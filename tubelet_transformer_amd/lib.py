"""ctypes binding of libtuber_hip.so (the C ABI declared in include/tuber_hip.h).

Signatures are parsed from the header, so Python, header and library cannot drift.
There is NO fallback: if the library is missing or a launcher returns non-zero this raises.
"""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "tuber_hip.h")
LIBPATH = os.path.join(HERE, "lib", "libtuber_hip.so")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "unsigned": ctypes.c_uint,
    "unsigned long long": ctypes.c_ulonglong, "hipStream_t": ctypes.c_void_p, "const char*": ctypes.c_char_p,
}
_lib = None
_sigs = None


def header_prototypes(path=HEADER):
    """[(ret, name, [(ctype_string, argname), ...])] for every `tuber_*` prototype in the header."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    protos = []
    for m in re.finditer(r"\b(int|long|const char\*)\s*(tuber_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        body = m.group(3).strip()
        if body and body != "void":
            for a in body.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                args.append((mm.group(1).strip(), mm.group(2)))
        protos.append((m.group(1), m.group(2), args))
    return protos


def _ctype(t):
    if "*" in t:
        return ctypes.c_void_p
    return _CT[t.replace("const ", "").strip()]


def load():
    global _lib, _sigs
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise RuntimeError(
            "libtuber_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The TubeR MI355X path has no CPU / eager fallback." % LIBPATH)
    lib = ctypes.CDLL(LIBPATH)
    sigs = {}
    for ret, name, args in header_prototypes():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype = _CT[ret]
        fn.argtypes = [_ctype(t) for t, _ in args]
        sigs[name] = args
    _lib, _sigs = lib, sigs
    if os.environ.get("TUBER_NT_WSK96"):             # measurement hook (DESIGN.md "Switches"): "0" = 64-row wave-split-K tiles everywhere
        lib.tuber_gemm_nt_wsk96_set(int(os.environ["TUBER_NT_WSK96"]))
    if os.environ.get("TUBER_NT_96"):                # measurement hook (round 6): "0" = 64-row tiles for the shapes that take 96-row tiles on the regular pipeline
        lib.tuber_gemm_nt_96_set(int(os.environ["TUBER_NT_96"]))
    return lib


def _conv(v, ctype_str):
    if "*" in ctype_str:
        if v is None:
            return None
        if isinstance(v, torch.Tensor):
            return v.data_ptr()
        if isinstance(v, (ctypes.Array, ctypes.Structure)):
            return ctypes.addressof(v)          # HOST argument blocks (tuber_gemm_tn_group)
        return v
    return v


def current_stream():
    return torch.cuda.current_stream().cuda_stream


_hook = None


def set_launch_hook(fn):
    """Install ``fn(name, args, launch)`` around every kernel launch (bench.py's HIP-event timing); None removes it."""
    global _hook
    _hook = fn


def call(name, *args):
    """Launch `name` on torch's current HIP stream (appended automatically when the prototype ends with it)."""
    if _hook is not None:
        return _hook(name, args, _call)
    return _call(name, *args)


def _call(name, *args):
    lib = load()
    sig = _sigs[name]
    if len(args) == len(sig) - 1 and sig and sig[-1][0] == "hipStream_t":
        args = args + (current_stream(),)
    if len(args) != len(sig):
        raise TypeError("%s expects %d args, got %d" % (name, len(sig), len(args)))
    rc = getattr(lib, name)(*[_conv(v, t) for v, (t, _) in zip(args, sig)])
    if rc != 0 and sig and sig[-1][0] == "hipStream_t":
        raise RuntimeError("%s failed with code %d" % (name, rc))
    return rc


def query(name, *args):
    """Helper calls without a stream (workspace sizing); returns the integer result."""
    lib = load()
    return getattr(lib, name)(*args)
