"""runmat_amd -- MI355X (gfx950) accelerate backend for RunMat's dense-array hot path.

Layout: `csrc/` holds the hand-written HIP kernels and the C-ABI library (librmhip.so,
include/rmhip.h); `provider.py` is the host-side mirror of the reference's `AccelProvider` trait;
`fusion.py` emits the WGSL requests the reference planner would send; `sharding.py` is the
one-process-per-GPU partitioning used by multi-GPU runs.
"""
from .provider import (GpuTensorHandle, HipProvider, ProviderError, ProviderLinsolveOptions, ProviderLinsolveResult,
                       ProviderLuResult, ReductionFlavor, wgsl_compile_check, wgsl_translate)

__all__ = ["GpuTensorHandle", "HipProvider", "ProviderError", "ProviderLinsolveOptions", "ProviderLinsolveResult",
           "ProviderLuResult", "ReductionFlavor", "wgsl_compile_check", "wgsl_translate"]
