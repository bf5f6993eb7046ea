"""ctypes binding of libecamp_hip.so (bfloat16 activations, the default) and libecamp_hip_f16.so (IEEE-half activations: the same sources
built with -DECAMP_HALF_F16, for `--amp fp16`).  The prototypes are parsed from include/ecamp_hip.h so the binding can never drift from
the declared C ABI.  There is NO fallback: if the library is missing or a call fails, we raise.

One process computes in ONE 16-bit format at a time: `set_half("f16")` makes every later call go to the f16 build (its option state,
plan caches and counters are its own); hip_ops.code() refuses a tensor whose dtype is not the active format."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# ECAMP_LIB (development A/B of two builds on one box, tools/ab_lib.sh): another build of the SAME library; the default is the in-tree one
LIB_PATH = os.environ.get("ECAMP_LIB") or os.path.join(HERE, "libecamp_hip.so")
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.environ.get("ECAMP_LIB_F16") or os.path.join(HERE, "libecamp_hip_f16.so")}
HEADER = os.path.join(os.path.dirname(HERE), "include", "ecamp_hip.h")

F32, BF16 = 0, 1

_CT = {
    "int": ctypes.c_int32, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float, "ecampStream_t": ctypes.c_void_p,
}


class EcampHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every `int ecamp_*(...)` / `int64_t ecamp_*` / `const char* ecamp_*` prototype."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(ecamp_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argl = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("", "void"):
                continue
            if "*" in a:
                ct, an = ctypes.c_void_p, a.split("*")[-1].strip()
            else:
                parts = a.split()
                ct, an = _CT[parts[-2]], parts[-1]
            argl.append((ct, an))
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int64 if ret == "int64_t" else ctypes.c_int32, argl)
    return protos


def abi_version_of_header(path=HEADER):
    m = re.search(r"#define\s+ECAMP_ABI_VERSION\s+(\d+)", open(path).read())
    if m is None:
        raise EcampHipError("include/ecamp_hip.h does not define ECAMP_ABI_VERSION")
    return int(m.group(1))


_libs = {}
_half = "bf16"
_protos = None


def half():
    """The active 16-bit activation format: "bf16" or "f16"."""
    return _half


def set_half(fmt):
    """Route every later call to the build whose 16-bit format is `fmt` ("bf16" / "f16", or torch.bfloat16 / torch.float16 /
    torch.float32 -- f32 leaves the choice alone); returns the previous format."""
    global _half
    name = {"torch.bfloat16": "bf16", "torch.float16": "f16", "torch.float32": _half}.get(str(fmt), fmt)
    if name not in LIB_PATHS:
        raise EcampHipError("unknown 16-bit format %r (bf16 or f16)" % (fmt,))
    prev, _half = _half, name
    return prev


def load(fmt=None):
    global _protos
    fmt = fmt or _half
    if fmt in _libs:
        return _libs[fmt]
    path = LIB_PATHS[fmt]
    if not os.path.exists(path):
        raise EcampHipError("%s not found at %s -- run `python -m ecamp_amd.build` (there is no CPU fallback)" % (os.path.basename(path), path))
    # torch first: it brings its own HIP runtime, and a process that loads /opt/rocm's runtime (through this library) before
    # torch's ends up with two runtimes of which the second sees no device ("no ROCm-capable device is detected")
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    _protos = parse_header()
    for name, (ret, args) in _protos.items():
        fn = getattr(lib, name)  # raises AttributeError if the symbol is not exported
        fn.restype = ret
        fn.argtypes = [a[0] for a in args]
    want = abi_version_of_header()
    got = lib.ecamp_abi_version()
    if got != want:
        raise EcampHipError("ABI version mismatch: %s was built for version %d, include/ecamp_hip.h declares %d -- rebuild "
                            "(`python -m ecamp_amd.build`); an older build must not be called with this header's argument lists"
                            % (path, got, want))
    if lib.ecamp_half_format() != (1 if fmt == "f16" else 0):
        raise EcampHipError("%s stores %s, not %s: the two builds were swapped (ECAMP_LIB / ECAMP_LIB_F16?)"
                            % (path, "IEEE half" if lib.ecamp_half_format() else "bfloat16", fmt))
    _libs[fmt] = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EcampHipError("%s failed (rc=%d): %s" % (name, rc, lib.ecamp_last_error().decode()))
