"""runmat_amd -- MI355X (gfx950) accelerate backend for RunMat's dense-array hot path.

Layout: `csrc/` holds the hand-written HIP kernels and the C-ABI library (librmhip.so,
include/rmhip.h); `provider.py` is the host-side mirror of the reference's `AccelProvider` trait;
`sharding.py` is the one-process-per-GPU partitioning used by multi-GPU runs.

NOT product code, kept here only because tests, `bench.py` and `sharding.py` import them: `fusion.py` and
`fusion_exec.py` are the REQUEST EMITTER - a restatement of what RunMat's own planner / executor (fusion.rs,
fusion_exec.rs; they stay in RunMat) put on the wire: the WGSL text and the call sequence around it.  The product is
`csrc/` (librmhip.so) behind `include/rmhip.h`; a RunMat build links that through `shim/hip_provider.rs` and never
loads these two modules.  They are checked against the reference generator's own unit tests
(tests/test_reference_kats.py::test_request_emitter_matches_the_generators_unit_tests).
"""
from .provider import (GpuTensorHandle, HipProvider, ProviderError, ProviderLinsolveOptions, ProviderLinsolveResult,
                       ProviderLuResult, ReductionFlavor, wgsl_compile_check, wgsl_translate)

__all__ = ["GpuTensorHandle", "HipProvider", "ProviderError", "ProviderLinsolveOptions", "ProviderLinsolveResult",
           "ProviderLuResult", "ReductionFlavor", "wgsl_compile_check", "wgsl_translate"]
