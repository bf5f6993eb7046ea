"""Invariants of the built gfx950 code objects (tools/check_isa.py): properties inline asm relies on and the compiler does not promise."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-objcopy", "clang-offload-bundler", "llvm-readelf")),
                    reason="ROCm LLVM binutils not present")
@pytest.mark.parametrize("name", ["libecamp_hip.so", "libecamp_hip_f16.so"])
def test_bias_registers_are_untouched_while_their_loads_travel(name):
    """gemm_q8.h: a tile's bias is requested by inline-asm vector loads that land behind the first K tile's counted DMA wait; in between no
    instruction of any persistent kernel may name the destination registers, and the statement must open with the s_nop that covers the
    VALU-written-SGPR -> VMEM hazard (an e4m3 kernel faulted without it).  Both builds of the sources -- bfloat16 and IEEE half -- are checked."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    lib = os.path.join(ROOT, "ecamp_amd", name)
    if not os.path.exists(lib):
        from ecamp_amd import build
        build.build(half="both")
    groups, problems = check_isa.check(lib)
    assert groups >= 17, "no bias-request groups found: the checker no longer recognises the code"
    assert not problems, "\n".join(problems)
