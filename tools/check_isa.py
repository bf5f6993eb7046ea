#!/usr/bin/env python3
"""Invariants of the built code objects that the sources cannot express (run by tests/test_isa.py on the in-tree library):

    python tools/check_isa.py [ecamp_amd/libecamp_hip.so]

1. gemm_q8.h (and gemm_q16.h in its 192-column form) request a tile's bias with eight (six) inline-asm `buffer_load_dwordx4` whose results land asynchronously; hipcc treats the
   outputs as defined at the asm statement.  Between the loads and the counted `s_waitcnt vmcnt` that lands them (the first K tile's
   DMA wait) NO instruction may name one of the 32 (24) destination registers -- a copy or a spill there would read them before the data
   arrived.  Checked for every persistent kernel that carries such a group, in program order up to the first wait that CAN land them:
   vector-memory operations retire in order, so a `vmcnt(N)` with N above the number of vector-memory operations between it and the
   group lands nothing of it and the scan continues behind it.  `python -m ecamp_amd.build` runs this check and fails on a problem.
2. Inside those asm statements a scalar register written by a VALU instruction (v_readlane: the kernels spill scalars) must not be read
   by a VMEM instruction within five wait states; the statement starts with `s_nop 4` for that reason -- checked: every group of
   asm bias loads is preceded by it.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LOAD = re.compile(r"buffer_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\], 0 offen( offset:\d+)?\s*(//|$)")


def code_objects(lib, tmp):
    """The gfx950 code objects of a linked library: .hip_fatbin holds one offload bundle per translation unit, back to back."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for n, a in enumerate(starts):
        part, co = os.path.join(tmp, "b%d.bin" % n), os.path.join(tmp, "b%d.co" % n)
        open(part, "wb").write(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co, "--unbundle"],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernels(co, pat):
    out = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", co], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if "FUNC" in l and re.search(pat, l)})


def regs_of(line):
    found = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", line.split("//")[0]):
        found.update(range(int(a), int(b) + 1) if a else [int(c)])
    return found


VMEM = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)(load|store|atomic)")


def vmcnt_of(line):
    """The vmcnt a wait instruction enforces, or None (a wait without a vmcnt field leaves the vector-memory counter alone)."""
    m = re.search(r"s_waitcnt\b(.*)", line.split("//")[0])
    if not m:
        return None
    v = re.search(r"vmcnt\((\d+)\)", m.group(1))
    return int(v.group(1)) if v else None


def check_kernel(co, name, problems):
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", co, "--disassemble-symbols=" + name], capture_output=True, text=True, check=True).stdout.splitlines()
    groups, i = 0, 0
    while i < len(dis):
        if not (LOAD.search(dis[i]) and " lds" not in dis[i]):
            i += 1
            continue
        j, dst = i, set()
        while j < len(dis) and LOAD.search(dis[j]) and " lds" not in dis[j]:
            m = LOAD.search(dis[j])
            dst.update(range(int(m.group(1)), int(m.group(2)) + 1))
            j += 1
        if (j - i, len(dst)) in ((8, 32), (6, 24)):   # a tile's bias request (eight-wave kernel: 8 loads; four-wave kernel, 192-column form: 6)
            groups += 1
            if "s_nop 4" not in dis[i - 1]:
                problems.append("%s: bias loads at line %d are not preceded by s_nop 4" % (name, i))
            # In program order up to the wait that can land the group.  Vector-memory operations retire in order, so `vmcnt(N)` lands the
            # bias loads only if at least N vector-memory operations were issued behind them: a wait whose count exceeds every
            # vector-memory operation that stands between it and the group cannot be the landing wait on ANY path -- the scan goes on
            # behind it (round 5 stopped at the first vmcnt of any count).  What this scan does not prove is which of the conditional DMA
            # issues between the group and the wait run on a given path; that invariant lives in the source (gemm_q8.h: the counted wait is
            # vmcnt(parts issued for tile t + 2) when another K tile follows and vmcnt(0) before the epilogue) and in the parity tests with
            # one- and two-K-tile shapes (tests/test_kernels_gpu.py: K = 136, 192, 200, 264).
            k, upper = j, 0
            while k < len(dis):
                n = vmcnt_of(dis[k])
                if n is not None and n <= upper:
                    break
                if regs_of(dis[k]) & dst:
                    problems.append("%s: line %d names a bias register in flight: %s" % (name, k, dis[k].split("//")[0].strip()))
                    break
                if VMEM.search(dis[k].split("//")[0]):
                    upper += 1
                k += 1
            else:
                problems.append("%s: bias loads at line %d: no wait behind them can land them" % (name, i))
        i = j
    return groups


def check(lib):
    problems, groups = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            for name in kernels(co, r"gemm_(bf16|f8)_q(8|16)_kernel"):
                groups += check_kernel(co, name, problems)
    return groups, problems


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ecamp_amd", "libecamp_hip.so")
    n, bad = check(lib)
    print("%d bias-request groups checked, %d problems" % (n, len(bad)))
    for b in bad:
        print("  " + b)
    sys.exit(1 if bad or n == 0 else 0)
