"""ecamp_amd -- MI355X-native implementation of ECAMP's pre-training hot path (ToniChopp/ECAMP).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every compute kernel is
hand-written HIP for gfx950 in `ecamp_amd/csrc`, reached through the C ABI declared in `include/ecamp_hip.h`.
The package mirrors the reference's `ECAMP/Pre-training` layout (`module/`, `util/`, `main_pretrain.py`).
"""
__version__ = "0.1.0"
