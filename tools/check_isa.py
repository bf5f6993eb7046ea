#!/usr/bin/env python3
"""Invariants of the built code objects that the sources cannot express (run by tests/test_isa.py on the in-tree library):

    python tools/check_isa.py [ecamp_amd/libecamp_hip.so]

1. gemm_q8.h (and gemm_q16.h in its 192-column form) request a tile's bias with eight (six) inline-asm `buffer_load_dwordx4` whose results land asynchronously; hipcc treats the
   outputs as defined at the asm statement.  Between the loads and the counted `s_waitcnt vmcnt` that lands them (the first K tile's
   DMA wait) NO instruction may name one of the 32 (24) destination registers -- a copy or a spill there would read them before the data
   arrived.  Checked for every persistent kernel that carries such a group.
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
            k = j
            while k < len(dis) and "s_waitcnt vmcnt" not in dis[k]:
                if regs_of(dis[k]) & dst:
                    problems.append("%s: line %d names a bias register in flight: %s" % (name, k, dis[k].split("//")[0].strip()))
                k += 1
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
