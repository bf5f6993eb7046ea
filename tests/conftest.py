import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---- the 16-bit activation format of a GPU test -----------------------------------------------------------------------------------
# libecamp_hip.so stores bfloat16, libecamp_hip_f16.so IEEE half (the reference's autocast format); one process calls one of them at a
# time (ecamp_amd._lib.set_half).  A test parametrised with `dtype` runs on the build of that dtype; a test that asks for the
# `both_halves` fixture runs once per build and reads its 16-bit torch dtype from h16(); everything else runs on the bfloat16 build.
def h16():
    import torch
    from ecamp_amd import _lib
    return torch.float16 if _lib.half() == "f16" else torch.bfloat16


@pytest.fixture(autouse=True)
def _half_format_of_the_test(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    from ecamp_amd import _lib
    cs = getattr(request.node, "callspec", None)
    dtype = cs.params.get("dtype") if cs is not None else None
    half = cs.params.get("both_halves") if cs is not None else None
    _lib.set_half("f16" if (dtype is torch.float16 or half == "f16") else "bf16")
    yield
    _lib.set_half("bf16")


@pytest.fixture(params=["bf16", "f16"])
def both_halves(request):
    return request.param
