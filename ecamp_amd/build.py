"""Build libecamp_hip.so (gfx950 only) from ecamp_amd/csrc/*.hip with hipcc.

    python -m ecamp_amd.build            # incremental
    python -m ecamp_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the authoring container; the resulting in-tree
`ecamp_amd/libecamp_hip.so` travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(os.path.dirname(HERE), "build")
LIB = os.path.join(HERE, "libecamp_hip.so")
LIB_F16 = os.path.join(HERE, "libecamp_hip_f16.so")   # the same sources with IEEE half as the 16-bit format (csrc/common.h)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-result"]


def _variant(half):
    if half == "f16":
        return os.path.join(OBJ, "f16"), LIB_F16, FLAGS + ["-DECAMP_HALF_F16=1"]
    return OBJ, LIB, FLAGS


def gemm_source_hash():
    """sha256 over the sources of the GEMM family (csrc/gemm.hip, gemm_q8.h, gemm_q16.h, gemm_args.h): the identity `profiles/rNN_pmc_traffic.json`
    is stamped with (tools/pmc_traffic.py) and bench.py checks before it quotes that file's bytes per GEMM launch as `roofline.traffic`
    -- a PMC pass of OTHER kernels must read as "not measured", not as last round's number."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_q8.h", "gemm_q16.h", "gemm_args.h"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, half="bf16"):
    """`half`: "bf16" -> libecamp_hip.so, "f16" -> libecamp_hip_f16.so (objects under build/f16), "both"."""
    if half == "both":
        lib = build(force, verbose, "bf16")
        build(force, verbose, "f16")
        return lib
    return _build_variant(force, verbose, *_variant(half))


def _build_variant(force, verbose, OBJ, LIB, FLAGS):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "ecamp_hip.h"))   # core.hip takes ECAMP_ABI_VERSION from it
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([hipcc] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        return src, r.returncode, r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, err in ex.map(cc, jobs):
                if verbose:
                    print("[ecamp_amd.build] hipcc %s -> rc %d" % (os.path.basename(src), rc), flush=True)
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s:\n%s" % (src, err))
                if verbose and "warning:" in err:  # e.g. "loop not unrolled": an accumulator array indexed at run time lives in scratch
                    print(err, flush=True)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("[ecamp_amd.build] linked %s" % LIB, flush=True)
        check_isa(verbose, LIB)
    return LIB


def check_isa(verbose=True, LIB=LIB):
    """The inline-asm invariants of the persistent GEMMs that hipcc does not promise (tools/check_isa.py), checked on every freshly
    linked library: a violation is a wrong-bias race, so it fails the build.  Without the ROCm binutils the check cannot run: said
    loudly, not silently (tests/test_isa.py then skips as well)."""
    tools = os.path.join(os.path.dirname(HERE), "tools")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(tools, "check_isa.py")):
        return
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objdump", "llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        print("[ecamp_amd.build] WARNING: ROCm LLVM binutils not found under %s -- the ISA invariants of the bias prefetch were NOT checked "
              "on this build" % llvm, file=sys.stderr, flush=True)
        return
    sys.path.insert(0, tools)
    try:
        import check_isa as ci
        groups, problems = ci.check(LIB)
    finally:
        sys.path.remove(tools)
    if problems or groups == 0:
        raise RuntimeError("ISA invariants violated in %s (%d bias-request groups):\n%s" % (LIB, groups, "\n".join(problems) or "no group recognised"))
    if verbose:
        print("[ecamp_amd.build] ISA invariants hold (%d bias-request groups)" % groups, flush=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv, half="f16" if "--f16" in sys.argv else "bf16" if "--bf16" in sys.argv else "both")
