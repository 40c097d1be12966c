"""runmat_amd -- MI355X (gfx950) accelerate backend for RunMat's dense-array hot path.

Layout: `csrc/` holds the hand-written HIP kernels and the C-ABI library (librmhip.so,
include/rmhip.h); `provider.py` is the host-side mirror of the reference's `AccelProvider` trait;
`sharding.py` is the one-process-per-GPU partitioning used by multi-GPU runs.

Nothing here generates requests: the WGSL text RunMat's planner puts on the wire (fusion.rs / fusion_exec.rs stay in RunMat) is
reproduced for tests and bench.py by the request emitter under tests/, outside the product package; the
sharded Monte-Carlo drivers take their two shaders as arguments.
"""
from .provider import (GpuTensorHandle, HipProvider, ProviderError, ProviderLinsolveOptions, ProviderLinsolveResult,
                       ProviderLuResult, ReduceDimResult, ReductionFlavor, wgsl_compile_check, wgsl_translate)

__all__ = ["GpuTensorHandle", "HipProvider", "ProviderError", "ProviderLinsolveOptions", "ProviderLinsolveResult",
           "ProviderLuResult", "ReduceDimResult", "ReductionFlavor", "wgsl_compile_check", "wgsl_translate"]
