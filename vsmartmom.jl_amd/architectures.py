"""Architecture dispatch surface, mirroring src/Architectures.jl:1-98 and the
method table a backend extension provides (ext/vSmartMOMCUDAExt.jl:21-27,46-57).

    CPU(), GPU()                 singleton architecture tags
    devi(arch)                   device handle (torch.device)
    array_type(arch)             converter: host array -> array living on `arch`
    architecture(array)          inverse lookup
    synchronize_if_gpu()         full-device sync (Architectures.jl:96)
    default_architecture()       GPU() when an MI355X is visible, else CPU()

`GPU()` here IS the MI355X/ROCm backend (key "Architectures.GPU()" in the
reference's ARCH_MAP, src/IO/Parameters.jl:80-88).  `CPU()` exists only so that
configs parse; this package ships no CPU compute path -- asking for RT on CPU()
raises, it never silently falls back.
"""
from __future__ import annotations

import numpy as np


class AbstractArchitecture:
    def __repr__(self):
        return "Architectures.%s()" % type(self).__name__


class CPU(AbstractArchitecture):
    pass


class GPU(AbstractArchitecture):
    def __init__(self, device_index: int = 0):
        self.device_index = device_index


ROCmGPU = GPU

ARCH_MAP = {
    "Architectures.CPU()": CPU,
    "Architectures.GPU()": GPU,
    "Architectures.ROCmGPU()": GPU,
    "CPU()": CPU,
    "GPU()": GPU,
}


def _torch():
    import torch
    return torch


def has_gpu() -> bool:
    torch = _torch()
    return bool(torch.cuda.is_available())


def default_architecture() -> AbstractArchitecture:
    return GPU() if has_gpu() else CPU()


def devi(arch: AbstractArchitecture):
    torch = _torch()
    if isinstance(arch, GPU):
        return torch.device("cuda", arch.device_index)
    return torch.device("cpu")


def array_type(arch: AbstractArchitecture):
    """Return a callable that moves a host array to `arch` (the reference's `arr_type(x)`)."""
    torch = _torch()
    dev = devi(arch)

    def convert(x):
        if isinstance(x, torch.Tensor):
            return x.to(dev)
        a = np.ascontiguousarray(x)
        if not a.flags.writeable:
            a = a.copy()
        return torch.as_tensor(a).to(dev)

    return convert


def architecture(array) -> AbstractArchitecture:
    torch = _torch()
    if isinstance(array, torch.Tensor) and array.is_cuda:
        return GPU(array.device.index or 0)
    return CPU()


def synchronize_if_gpu():
    torch = _torch()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def to_host(x) -> np.ndarray:
    """`Array(x)` of the reference: device -> host."""
    torch = _torch()
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)
