"""vsmartmom.jl_amd -- MI355X (gfx950) engine for vSmartMOM.jl's `rt_run` CoreRT hot path.

The directory name carries a dot (it mirrors the reference's name), so it cannot
be imported with a plain `import` statement; use the shim at the repo root:

    import vsmartmom_jl_amd as vsm
    vsm.CoreRT.rt_run(model)

Sub-modules
    Architectures   CPU()/GPU() dispatch surface (src/Architectures.jl)
    CoreRT          batched_mul / batch_inv_ / elemental_ / doubling_ / interaction_ / rt_kernel_ / rt_run
    host_model      host-side producers of the hot path's inputs (streams, Z moments, layer mixing)
    CoreRTLin       linearized (Jacobian) pass: rt_run(model, lin_model, ...)
    CoreRTRaman     rotational-Raman pass: rt_run(RS_type::RRS, model, iBand)
    parallel        spectral-axis sharding over the GPUs of one node (torch.distributed / RCCL)
"""
from . import _lib, host_model  # noqa: F401
from . import architectures as Architectures  # noqa: F401
from . import core_rt as CoreRT  # noqa: F401
from . import parallel  # noqa: F401
from . import core_rt_lin as CoreRTLin  # noqa: F401
from . import core_rt_raman as CoreRTRaman  # noqa: F401
from . import raman_inputs  # noqa: F401
from . import io_yaml  # noqa: F401
from ._lib import VSMError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
