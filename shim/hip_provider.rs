//! hip_provider.rs -- the binding a RunMat maintainer adds to plug librmhip.so in as a third
//! `AccelProvider` backend (next to `WgpuProvider` and `InProcessProvider`).
//!
//! NOT compiled in this repository (the build image has no Rust toolchain).  Drop it, together with the generated
//! `rmhip_sys.rs` (the FFI declarations, derived from `include/rmhip.h`), into
//! `crates/runmat-accelerate/src/backend/hip/`, add `links = "rmhip"` / a build.rs that
//! emits `cargo:rustc-link-lib=dylib=rmhip`, and call `register_hip_provider()` before
//! `initialize_acceleration_provider_with` (which returns early when a provider is already
//! registered, crates/runmat-accelerate/src/lib.rs:179-181).
//!
//! Every method maps 1:1 onto one C entry point; any non-zero status becomes `Err(anyhow!(..))`,
//! which RunMat's callers already treat as "fall back to the CPU builtin"
//! (mtimes.rs:212-216, mldivide.rs:223-229, runner.rs:1140-1142).

use anyhow::{anyhow, Result};
use runmat_accelerate_api::{
    AccelProvider, AccelProviderFuture, ApiDeviceInfo, CorrcoefNormalization, CorrcoefOptions, CorrcoefRows, CovNormalization, CovRows, CovarianceOptions, FindDirection, GpuTensorHandle, GpuTensorStorage,
    HostLogicalOwned, HostTensorOwned, HostTensorView, IsMemberOptions, IsMemberResult, SetdiffOptions, SetdiffOrder, SetdiffResult, UnionOptions, UnionOrder, UnionResult, UniqueOccurrence, UniqueOptions, UniqueOrder, UniqueResult, ImageNormalizeDescriptor, ImfilterMode, ImfilterOptions, ImfilterPadding, ImfilterShape, KernelAttrTelemetry, MeshgridAxisView, ProviderMeshgridResult, ProviderPolyderQuotient, ProviderPolyvalOptions, KernelLaunchTelemetry, MatmulEpilogue,
    PowerStepEpilogue, ProviderBandwidth, ProviderCovarianceToCorrelationResult, ProviderHilbertRequest, ProviderCondNorm, ProviderPinvOptions, ProviderIirFilterOptions, ProviderIirFilterResult, ProviderInterp1Extrapolation, ProviderInterp1Method, ProviderInterp1Request, ProviderConv1dOptions, ProviderConvMode, ProviderConvOrientation, ProviderCholResult, ProviderCummaxResult, ProviderCumminResult, ProviderDispatchStats, ProviderInvOptions, ProviderFallbackStat, ProviderFindResult, ProviderHermitianKind, ProviderLinsolveOptions,
    ProviderLinsolveResult, ProviderLuResult, ProviderMoments2, ProviderMovingWindowEndpoints, ProviderMovingWindowOp, ProviderMovingWindowRequest, ProviderNanMode, ProviderNdgridRequest, ProviderNormOrder, ProviderNdgridResult, ProviderPrecision, ProviderScanDirection,
    ProviderStdNormalization, ProviderSymmetryKind, ProviderTelemetry, ProviderTrapezoidSpacing, ReduceDimResult, ReductionFlavor, ScaleOp, SortComparison, SortOrder, SortResult, SortRowsColumnSpec,
};
use std::ffi::{c_char, c_int, c_void, CStr, CString};

// The raw FFI surface - `#[repr(C)]` structs, op-code constants and the `extern "C"` block - is GENERATED from include/rmhip.h
// by scripts/gen_bindings.py (shim/rmhip_sys.rs); tests/test_bindings.py fails when it is stale or when a trait method an
// `@serves` tag of the header names has no implementation below.
#[path = "rmhip_sys.rs"]
mod sys;
use sys::*;

pub struct HipProvider {
    ctx: *mut RmhipCtx,
    precision: ProviderPrecision,
    device_id: u32,
}
// `AccelProvider: Send + Sync` (lib.rs:1386).  The context pointer is only ever handed to librmhip entry points, and
// every entry point takes the context's call mutex first (`Context::call_mu`, taken by CTX_OR_FAIL in
// runmat_amd/csrc/rmhip_core.cpp / rmhip_ops.cpp): overlapping calls from several host threads serialise inside the
// library instead of interleaving on its per-context state (scratch, RNG state, the look-ahead LU's stream
// retargeting).  tests/test_gpu_threads.py hammers one context from two threads.
unsafe impl Send for HipProvider {}
unsafe impl Sync for HipProvider {}

fn check(rc: c_int) -> Result<()> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(rmhip_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("rmhip[{rc}]: {msg}"))
}

impl HipProvider {
    fn real_part(&self, part: c_int, a: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_real_part(self.ctx, part, self.own(a)?, &mut out) })?;
        self.handle(out)
    }
    // trapz_dim / cumtrapz_dim: ProviderTrapezoidSpacing (lib.rs:1060-1066) -> (kind, scalar, handle) of rmhip_trapz_dim
    fn trapezoid(&self, input: &GpuTensorHandle, dim: usize, spacing: ProviderTrapezoidSpacing<'_>, cumulative: c_int) -> Result<GpuTensorHandle> {
        let (kind, scalar, handle) = match spacing {
            ProviderTrapezoidSpacing::Unit => (0, 0.0, 0u64),
            ProviderTrapezoidSpacing::Scalar(v) => (1, v, 0u64),
            ProviderTrapezoidSpacing::ScalarHandle(h) => (2, 0.0, self.own(h)?),
            ProviderTrapezoidSpacing::Vector(h) => (3, 0.0, self.own(h)?),
            ProviderTrapezoidSpacing::Tensor(h) => (4, 0.0, self.own(h)?),
        };
        let mut out = 0u64;
        check(unsafe { rmhip_trapz_dim(self.ctx, self.own(input)?, dim as c_int, cumulative, kind, scalar, handle, &mut out) })?;
        self.handle(out)
    }

    pub fn new(device_ordinal: i32) -> Result<Self> {
        Self::with_precision(device_ordinal, ProviderPrecision::F64)
    }
    /// F32: tensors live in HBM as f32 (the planner then emits f32 shaders, fusion.rs:1525); host views stay f64.
    pub fn with_precision(device_ordinal: i32, precision: ProviderPrecision) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { rmhip_init(device_ordinal, &mut ctx) })?;
        if matches!(precision, ProviderPrecision::F32) {
            check(unsafe { rmhip_set_precision(ctx, 32) })?;
        }
        Ok(Self { ctx, precision, device_id: runmat_accelerate_api::next_device_id() }) // lib.rs:3279
    }
    fn handle(&self, id: u64) -> Result<GpuTensorHandle> {
        let mut rank = 16usize;
        let mut shape = [0usize; 16];
        check(unsafe { rmhip_shape(self.ctx, id, &mut rank, shape.as_mut_ptr()) })?;
        Ok(GpuTensorHandle { shape: shape[..rank].to_vec(), device_id: self.device_id, buffer_id: id })
    }
    fn window(&self, kind: c_int, len: usize, periodic: bool) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_window(self.ctx, kind, len, periodic as c_int, &mut out) })?;
        self.handle(out)
    }
    fn complex_handle(&self, id: u64) -> Result<GpuTensorHandle> {
        let h = self.handle(id)?;
        runmat_accelerate_api::set_handle_storage(&h, GpuTensorStorage::ComplexInterleaved);
        Ok(h)
    }
    fn transform(&self, handle: &GpuTensorHandle, len: Option<usize>, dim: usize, inverse: c_int) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        let len = len.map(|l| l as i64).unwrap_or(-1);
        check(unsafe { rmhip_fft_dim(self.ctx, self.own(handle)?, len, dim as c_int, inverse, &mut out) })?;
        self.complex_handle(out)
    }
    fn complex(&self, real: &GpuTensorHandle, imag: Option<&GpuTensorHandle>) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        let im = match imag { Some(h) => self.own(h)?, None => 0 };
        check(unsafe { rmhip_complex(self.ctx, self.own(real)?, im, &mut out) })?;
        self.complex_handle(out)
    }
    fn own(&self, h: &GpuTensorHandle) -> Result<u64> {
        if h.device_id != self.device_id {
            return Err(anyhow!("handle belongs to device {}", h.device_id)); // io.rs:269-275
        }
        Ok(h.buffer_id)
    }
    fn unary(&self, op: c_int, a: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_unary(self.ctx, op, self.own(a)?, &mut out) })?;
        self.handle(out)
    }
    fn binary(&self, op: c_int, a: &GpuTensorHandle, b: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_binary(self.ctx, op, self.own(a)?, self.own(b)?, &mut out) })?;
        self.handle(out)
    }
}

impl Drop for HipProvider {
    fn drop(&mut self) {
        unsafe { rmhip_shutdown(self.ctx) };
    }
}

// One trait method per line: the hooks differ only in the op code handed to the library.
fn conv_mode(mode: ProviderConvMode) -> c_int {
    match mode { ProviderConvMode::Full => 0, ProviderConvMode::Same => 1, ProviderConvMode::Valid => 2 }
}

macro_rules! unary_hooks { ($($name:ident => $op:expr),* $(,)?) => { $(
    fn $name<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.unary($op, a) }) }
)* } }
macro_rules! binary_hooks { ($($name:ident => $op:expr),* $(,)?) => { $(
    fn $name<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.binary($op, a, b) }) }
)* } }
macro_rules! logical_hooks { ($($name:ident => $op:expr),* $(,)?) => { $(
    fn $name(&self, a: &GpuTensorHandle, b: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.binary($op, a, b) }
)* } }
macro_rules! scalar_hooks { ($($name:ident => $op:expr),* $(,)?) => { $(
    fn $name(&self, a: &GpuTensorHandle, scalar: f64) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_scalar(self.ctx, $op, self.own(a)?, scalar, &mut out) })?;
        self.handle(out)
    }
)* } }
macro_rules! reduce_all_hooks { ($($name:ident => $op:expr),* $(,)?) => { $(
    fn $name<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce(self.ctx, $op, self.own(a)?, -1, 0, &mut out) })?;
            self.handle(out)
        })
    }
)* } }
macro_rules! reduce_dim_hooks { ($($name:ident => $op:expr),* $(,)?) => { $(
    fn $name<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce(self.ctx, $op, self.own(a)?, dim as c_int, 0, &mut out) })?;
            self.handle(out)
        })
    }
)* } }

impl AccelProvider for HipProvider {
    fn upload(&self, host: &HostTensorView) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_upload(self.ctx, host.data.as_ptr(), host.shape.as_ptr(), host.shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: host.shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn download<'a>(&'a self, h: &'a GpuTensorHandle) -> AccelProviderFuture<'a, HostTensorOwned> {
        Box::pin(async move {
            let mut cplx: c_int = 0;
            check(unsafe { rmhip_storage(self.ctx, self.own(h)?, &mut cplx) })?;
            let storage = if cplx != 0 { GpuTensorStorage::ComplexInterleaved } else { GpuTensorStorage::Real };
            let n: usize = h.shape.iter().product::<usize>() * if cplx != 0 { 2 } else { 1 };
            let mut data = vec![0.0f64; n];
            check(unsafe { rmhip_download(self.ctx, self.own(h)?, data.as_mut_ptr(), n) })?;
            Ok(HostTensorOwned { data, shape: h.shape.clone(), storage })
        })
    }
    fn free(&self, h: &GpuTensorHandle) -> Result<()> {
        check(unsafe { rmhip_free(self.ctx, self.own(h)?) })
    }
    fn device_id(&self) -> u32 { self.device_id }
    fn precision(&self) -> ProviderPrecision { self.precision }

    fn fused_elementwise(&self, shader: &str, inputs: &[GpuTensorHandle], output_shape: &[usize], len: usize)
        -> Result<GpuTensorHandle> {
        Ok(self.fused_elementwise_multi(shader, inputs, output_shape, len, 1)?.remove(0))
    }
    fn fused_elementwise_multi(&self, shader: &str, inputs: &[GpuTensorHandle], output_shape: &[usize], len: usize,
        num_outputs: usize) -> Result<Vec<GpuTensorHandle>> {
        let src = CString::new(shader)?;
        let ids = inputs.iter().map(|h| self.own(h)).collect::<Result<Vec<_>>>()?;
        let mut outs = vec![0u64; num_outputs];
        check(unsafe { rmhip_fused_elementwise(self.ctx, src.as_ptr(), ids.as_ptr(), ids.len(), output_shape.as_ptr(),
            output_shape.len(), len, num_outputs, outs.as_mut_ptr()) })?;
        Ok(outs.into_iter().map(|id| GpuTensorHandle { shape: output_shape.to_vec(), device_id: self.device_id, buffer_id: id }).collect())
    }
    fn fused_reduction(&self, shader: &str, inputs: &[GpuTensorHandle], output_shape: &[usize], reduce_len: usize,
        num_slices: usize, workgroup_size: u32, flavor: ReductionFlavor) -> Result<GpuTensorHandle> {
        let src = CString::new(shader)?;
        let ids = inputs.iter().map(|h| self.own(h)).collect::<Result<Vec<_>>>()?;
        let (code, scale) = match flavor {
            ReductionFlavor::Sum => (0, 1.0),
            ReductionFlavor::Mean => (1, 1.0), // the library divides by the count like the CPU path
            ReductionFlavor::CustomScale(s) => (2, s),
        };
        let mut out = 0u64;
        check(unsafe { rmhip_fused_reduction(self.ctx, src.as_ptr(), ids.as_ptr(), ids.len(), output_shape.as_ptr(),
            output_shape.len(), reduce_len, num_slices, workgroup_size, code, scale, &mut out) })?;
        Ok(GpuTensorHandle { shape: output_shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }

    // per-op hooks (lib.rs:1890-2355): one line per trait method, generated by the macros above the impl
    unary_hooks! {
        unary_sin => RMHIP_SIN, unary_cos => RMHIP_COS, unary_tan => RMHIP_TAN, unary_asin => RMHIP_ASIN, unary_acos => RMHIP_ACOS,
        unary_atan => RMHIP_ATAN, unary_sinh => RMHIP_SINH, unary_cosh => RMHIP_COSH, unary_tanh => RMHIP_TANH,
        unary_asinh => RMHIP_ASINH, unary_acosh => RMHIP_ACOSH, unary_atanh => RMHIP_ATANH, unary_exp => RMHIP_EXP,
        unary_expm1 => RMHIP_EXPM1, unary_log => RMHIP_LOG, unary_log2 => RMHIP_LOG2, unary_log10 => RMHIP_LOG10,
        unary_log1p => RMHIP_LOG1P, unary_sqrt => RMHIP_SQRT, unary_abs => RMHIP_ABS, unary_sign => RMHIP_SIGN,
        unary_floor => RMHIP_FLOOR, unary_ceil => RMHIP_CEIL, unary_round => RMHIP_ROUND, unary_fix => RMHIP_FIX,
        unary_pow2 => RMHIP_EXP2, unary_heaviside => RMHIP_HEAVISIDE, unary_single => RMHIP_SINGLE, unary_double => RMHIP_DOUBLE,
        unary_erf => RMHIP_ERF, unary_sinc => RMHIP_SINC, unary_gamma => RMHIP_GAMMA, unary_factorial => RMHIP_FACTORIAL,
        unary_nextpow2 => RMHIP_NEXTPOW2, unary_gammaln => RMHIP_GAMMALN, unary_erfcinv => RMHIP_ERFCINV,
    }
    binary_hooks! {
        elem_add => RMHIP_ADD, elem_sub => RMHIP_SUB, elem_mul => RMHIP_MUL, elem_div => RMHIP_DIV, elem_pow => RMHIP_POW,
        elem_max => RMHIP_MAX, elem_min => RMHIP_MIN, elem_hypot => RMHIP_HYPOT, elem_atan2 => RMHIP_ATAN2,
        elem_eq => RMHIP_EQ, elem_ne => RMHIP_NE, elem_lt => RMHIP_LT, elem_le => RMHIP_LE, elem_gt => RMHIP_GT, elem_ge => RMHIP_GE,
    }
    logical_hooks! { logical_and => RMHIP_AND, logical_or => RMHIP_OR, logical_xor => RMHIP_XOR }
    fn logical_not(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.unary(RMHIP_NOT, a) }
    fn logical_isnan(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.unary(RMHIP_ISNAN, a) }
    fn logical_isinf(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.unary(RMHIP_ISINF, a) }
    fn logical_isfinite(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.unary(RMHIP_ISFINITE, a) }
    scalar_hooks! {
        scalar_add => RMHIP_SADD, scalar_sub => RMHIP_SSUB, scalar_mul => RMHIP_SMUL, scalar_div => RMHIP_SDIV,
        scalar_rsub => RMHIP_SRSUB, scalar_rdiv => RMHIP_SRDIV, scalar_max => RMHIP_SMAX, scalar_min => RMHIP_SMIN,
    }
    reduce_all_hooks! { reduce_mean => RMHIP_RMEAN, reduce_min => RMHIP_RMIN, reduce_max => RMHIP_RMAX, reduce_prod => RMHIP_RPROD }
    reduce_dim_hooks! { reduce_mean_dim => RMHIP_RMEAN, reduce_prod_dim => RMHIP_RPROD }
    fn dot<'a>(&'a self, lhs: &'a GpuTensorHandle, rhs: &'a GpuTensorHandle, dim: Option<usize>) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            let d = dim.map(|d| d as c_int).unwrap_or(-1); // None: first non-singleton dimension
            check(unsafe { rmhip_dot(self.ctx, self.own(lhs)?, self.own(rhs)?, d, &mut out) })?;
            self.handle(out)
        })
    }
    // The library keeps each buffer's shape, so the trait default (edit the handle only, lib.rs:2676-2684) is not
    // enough; like the wgpu provider's reshape_exec the library updates the entry in place and hands back the SAME
    // buffer id (`out == handle.buffer_id`), so the consumed source handle needs no separate free.
    fn reshape(&self, handle: &GpuTensorHandle, new_shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_reshape(self.ctx, self.own(handle)?, new_shape.as_ptr(), new_shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: new_shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn zeros(&self, shape: &[usize]) -> Result<GpuTensorHandle> { self.fill(shape, 0.0) }
    fn ones(&self, shape: &[usize]) -> Result<GpuTensorHandle> { self.fill(shape, 1.0) }
    fn fill(&self, shape: &[usize], value: f64) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_fill(self.ctx, value, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }

    fn reduce_sum<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce(self.ctx, 0, self.own(a)?, -1, 0, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_sum_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce(self.ctx, 0, self.own(a)?, dim as c_int, 0, &mut out) })?;
            self.handle(out)
        })
    }
    // reduce_min_dim / reduce_max_dim -> ReduceDimResult { values, indices } (lib.rs:2864-2883): the runtime calls these in
    // "includenan" mode only (min.rs:795-800) and expects its host path's answer: first occurrence, first NaN wins
    fn reduce_min_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize) -> AccelProviderFuture<'a, ReduceDimResult> {
        Box::pin(async move {
            let (mut v, mut i) = (0u64, 0u64);
            check(unsafe { rmhip_reduce_minmax_dim(self.ctx, 2 /* RMHIP_RMIN */, self.own(a)?, dim as c_int, 0, &mut v, &mut i) })?;
            Ok(ReduceDimResult { values: self.handle(v)?, indices: self.handle(i)? })
        })
    }
    fn reduce_max_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize) -> AccelProviderFuture<'a, ReduceDimResult> {
        Box::pin(async move {
            let (mut v, mut i) = (0u64, 0u64);
            check(unsafe { rmhip_reduce_minmax_dim(self.ctx, 3 /* RMHIP_RMAX */, self.own(a)?, dim as c_int, 0, &mut v, &mut i) })?;
            Ok(ReduceDimResult { values: self.handle(v)?, indices: self.handle(i)? })
        })
    }
    fn reduce_std_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize, normalization: ProviderStdNormalization, nan_mode: ProviderNanMode) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            let norm = matches!(normalization, ProviderStdNormalization::Population) as c_int;
            let nan = matches!(nan_mode, ProviderNanMode::Omit) as c_int;
            check(unsafe { rmhip_reduce_std(self.ctx, self.own(a)?, dim as c_int, norm, nan, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_std<'a>(&'a self, a: &'a GpuTensorHandle, normalization: ProviderStdNormalization, nan_mode: ProviderNanMode) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            let norm = matches!(normalization, ProviderStdNormalization::Population) as c_int;
            let nan = matches!(nan_mode, ProviderNanMode::Omit) as c_int;
            check(unsafe { rmhip_reduce_std(self.ctx, self.own(a)?, -1, norm, nan, &mut out) })?;
            self.handle(out)
        })
    }
    // reduce_nnz(_dim) / reduce_any(_dim) / reduce_all(_dim) (lib.rs:2730-2742, 2803-2850): op 0 / 1 / 2 of rmhip_reduce_truth; the
    // forms without `_dim` are the same calls with dim = -1
    fn reduce_nnz_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce_truth(self.ctx, 0, self.own(a)?, dim as c_int, 0, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_any_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize, omit_nan: bool) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce_truth(self.ctx, 1, self.own(a)?, dim as c_int, omit_nan as c_int, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_all_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize, omit_nan: bool) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce_truth(self.ctx, 2, self.own(a)?, dim as c_int, omit_nan as c_int, &mut out) })?;
            self.handle(out)
        })
    }
    fn cumsum_scan(&self, input: &GpuTensorHandle, dim: usize, direction: ProviderScanDirection, nan_mode: ProviderNanMode) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        let rev = matches!(direction, ProviderScanDirection::Reverse) as c_int;
        let nan = matches!(nan_mode, ProviderNanMode::Omit) as c_int;
        check(unsafe { rmhip_cumulative(self.ctx, 0, self.own(input)?, dim as c_int, rev, nan, &mut out) })?;
        self.handle(out)
    }
    fn cumprod_scan(&self, input: &GpuTensorHandle, dim: usize, direction: ProviderScanDirection, nan_mode: ProviderNanMode) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        let rev = matches!(direction, ProviderScanDirection::Reverse) as c_int;
        let nan = matches!(nan_mode, ProviderNanMode::Omit) as c_int;
        check(unsafe { rmhip_cumulative(self.ctx, 1, self.own(input)?, dim as c_int, rev, nan, &mut out) })?;
        self.handle(out)
    }
    // cummin_scan / cummax_scan -> ProviderCumminResult { values, indices } (lib.rs:2918-2935; ProviderCummaxResult is its alias)
    fn cummin_scan(&self, input: &GpuTensorHandle, dim: usize, direction: ProviderScanDirection, nan_mode: ProviderNanMode) -> Result<ProviderCumminResult> {
        let (mut v, mut i) = (0u64, 0u64);
        let rev = matches!(direction, ProviderScanDirection::Reverse) as c_int;
        let nan = matches!(nan_mode, ProviderNanMode::Omit) as c_int;
        check(unsafe { rmhip_cumextreme(self.ctx, 0, self.own(input)?, dim as c_int, rev, nan, &mut v, &mut i) })?;
        Ok(ProviderCumminResult { values: self.handle(v)?, indices: self.handle(i)? })
    }
    fn cummax_scan(&self, input: &GpuTensorHandle, dim: usize, direction: ProviderScanDirection, nan_mode: ProviderNanMode) -> Result<ProviderCummaxResult> {
        let (mut v, mut i) = (0u64, 0u64);
        let rev = matches!(direction, ProviderScanDirection::Reverse) as c_int;
        let nan = matches!(nan_mode, ProviderNanMode::Omit) as c_int;
        check(unsafe { rmhip_cumextreme(self.ctx, 1, self.own(input)?, dim as c_int, rev, nan, &mut v, &mut i) })?;
        Ok(ProviderCummaxResult { values: self.handle(v)?, indices: self.handle(i)? })
    }
    // diff_dim (lib.rs:2596-2603): the reference's own output order (column_major = 0), the shape diff_tensor_host reports
    fn diff_dim(&self, handle: &GpuTensorHandle, order: usize, dim: usize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_diff_dim(self.ctx, self.own(handle)?, order, dim as c_int, 0, &mut out) })?;
        self.handle(out)
    }
    // sort_dim -> SortResult with HOST tensors (lib.rs:2358-2366, 1085-1088): both device results are downloaded and freed
    fn sort_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize, order: SortOrder, comparison: SortComparison) -> AccelProviderFuture<'a, SortResult> {
        Box::pin(async move {
            let (mut v, mut i) = (0u64, 0u64);
            let desc = matches!(order, SortOrder::Descend) as c_int;
            let abs = matches!(comparison, SortComparison::Abs) as c_int;
            check(unsafe { rmhip_sort_dim(self.ctx, self.own(a)?, dim as c_int, desc, abs, &mut v, &mut i) })?;
            let (hv, hi) = (self.handle(v)?, self.handle(i)?);
            let values = self.download(&hv).await;
            let indices = self.download(&hi).await;
            self.free(&hv)?;
            self.free(&hi)?;
            Ok(SortResult { values: values?, indices: indices? })
        })
    }
    fn sort_rows<'a>(&'a self, a: &'a GpuTensorHandle, columns: &'a [SortRowsColumnSpec], comparison: SortComparison) -> AccelProviderFuture<'a, SortResult> {
        Box::pin(async move {
            let idx: Vec<usize> = columns.iter().map(|s| s.index).collect();
            let desc: Vec<c_int> = columns.iter().map(|s| matches!(s.order, SortOrder::Descend) as c_int).collect();
            let abs = matches!(comparison, SortComparison::Abs) as c_int;
            let (mut v, mut i) = (0u64, 0u64);
            check(unsafe { rmhip_sort_rows(self.ctx, self.own(a)?, idx.as_ptr(), desc.as_ptr(), columns.len(), abs, &mut v, &mut i) })?;
            let (hv, hi) = (self.handle(v)?, self.handle(i)?);
            let values = self.download(&hv).await;
            let indices = self.download(&hi).await;
            self.free(&hv)?;
            self.free(&hi)?;
            Ok(SortResult { values: values?, indices: indices? })
        })
    }
    // find -> ProviderFindResult { linear, rows, cols, values: Some(..) } (lib.rs:2937-2944, 623-628)
    fn find(&self, a: &GpuTensorHandle, limit: Option<usize>, direction: FindDirection) -> Result<ProviderFindResult> {
        let (mut l, mut r, mut c, mut v) = (0u64, 0u64, 0u64, 0u64);
        let last = matches!(direction, FindDirection::Last) as c_int;
        let lim = limit.map(|k| k as i64).unwrap_or(-1);
        check(unsafe { rmhip_find(self.ctx, self.own(a)?, lim, last, &mut l, &mut r, &mut c, &mut v) })?;
        Ok(ProviderFindResult { linear: self.handle(l)?, rows: self.handle(r)?, cols: self.handle(c)?, values: Some(self.handle(v)?) })
    }
    // reduce_median: every element as one line; reduce_median_dim: include-NaN median along the zero-based dim (lib.rs:2833-2845)
    fn reduce_median<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce_median(self.ctx, self.own(a)?, -1, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_median_dim<'a>(&'a self, a: &'a GpuTensorHandle, dim: usize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce_median(self.ctx, self.own(a)?, dim as c_int, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_mean_nd<'a>(&'a self, a: &'a GpuTensorHandle, dims_zero_based: &'a [usize]) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_reduce_nd(self.ctx, 1 /* RMHIP_RMEAN */, self.own(a)?, dims_zero_based.as_ptr(), dims_zero_based.len(), 0, &mut out) })?;
            self.handle(out)
        })
    }
    fn reduce_moments_nd<'a>(&'a self, a: &'a GpuTensorHandle, dims_zero_based: &'a [usize]) -> AccelProviderFuture<'a, ProviderMoments2> {
        Box::pin(async move {
            let (mut mean, mut ex2) = (0u64, 0u64);
            check(unsafe { rmhip_reduce_moments_nd(self.ctx, self.own(a)?, dims_zero_based.as_ptr(), dims_zero_based.len(), &mut mean, &mut ex2) })?;
            Ok(ProviderMoments2 { mean: self.handle(mean)?, ex2: self.handle(ex2)? })
        })
    }
    fn matmul<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_matmul(self.ctx, self.own(a)?, self.own(b)?, &mut out) })?;
            self.handle(out)
        })
    }
    fn mldivide<'a>(&'a self, lhs: &'a GpuTensorHandle, rhs: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_mldivide(self.ctx, self.own(lhs)?, self.own(rhs)?, &mut out) })?;
            self.handle(out)
        })
    }
    // chol: the device's success path (info = 0); Err for a matrix that is not symmetric positive definite -> chol.rs:331-342 host path
    fn chol<'a>(&'a self, a: &'a GpuTensorHandle, lower: bool) -> AccelProviderFuture<'a, ProviderCholResult> {
        Box::pin(async move {
            let (mut out, mut info) = (0u64, 0u32);
            check(unsafe { rmhip_chol(self.ctx, self.own(a)?, lower as c_int, &mut out, &mut info) })?;
            Ok(ProviderCholResult { factor: self.handle(out)?, info })
        })
    }
    // inv(A) = A \ I on the LU path; Err (singular, non-square) sends inv.rs back to its host code, which words the error
    fn inv<'a>(&'a self, matrix: &'a GpuTensorHandle, _options: ProviderInvOptions) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_inv(self.ctx, self.own(matrix)?, &mut out) })?;
            self.handle(out)
        })
    }
    fn mrdivide<'a>(&'a self, lhs: &'a GpuTensorHandle, rhs: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_mrdivide(self.ctx, self.own(lhs)?, self.own(rhs)?, &mut out) })?;
            self.handle(out)
        })
    }
    fn linsolve<'a>(&'a self, lhs: &'a GpuTensorHandle, rhs: &'a GpuTensorHandle, o: &'a ProviderLinsolveOptions)
        -> AccelProviderFuture<'a, ProviderLinsolveResult> {
        Box::pin(async move {
            let c = RmhipLinsolveOptions { lower: o.lower as c_int, upper: o.upper as c_int, rectangular: o.rectangular as c_int,
                transposed: o.transposed as c_int, conjugate: o.conjugate as c_int, symmetric: o.symmetric as c_int,
                posdef: o.posdef as c_int, need_rcond: o.need_rcond as c_int, has_rcond: o.rcond.is_some() as c_int,
                rcond: o.rcond.unwrap_or(0.0) };
            let (mut out, mut rcond) = (0u64, f64::NAN);
            check(unsafe { rmhip_linsolve(self.ctx, self.own(lhs)?, self.own(rhs)?, &c, &mut out, &mut rcond) })?;
            Ok(ProviderLinsolveResult { solution: self.handle(out)?, reciprocal_condition: rcond })
        })
    }
    fn matmul_power_step<'a>(&'a self, lhs: &'a GpuTensorHandle, rhs: &'a GpuTensorHandle, ep: &'a PowerStepEpilogue)
        -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_matmul_power_step(self.ctx, self.own(lhs)?, self.own(rhs)?, ep.epsilon, &mut out) })?;
            self.handle(out)
        })
    }
    fn image_normalize<'a>(&'a self, input: &'a GpuTensorHandle, d: &'a ImageNormalizeDescriptor) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let c = RmhipImageNormalize { batch: d.batch, height: d.height, width: d.width, epsilon: d.epsilon,
                has_gain: d.gain.is_some() as c_int, has_bias: d.bias.is_some() as c_int, has_gamma: d.gamma.is_some() as c_int,
                clamp_zero: d.clamp_zero as c_int, gain: d.gain.unwrap_or(0.0), bias: d.bias.unwrap_or(0.0), gamma: d.gamma.unwrap_or(0.0) };
            let mut out = 0u64;
            check(unsafe { rmhip_image_normalize(self.ctx, self.own(input)?, &c, &mut out) })?;
            self.handle(out)
        })
    }
    fn covariance<'a>(&'a self, matrix: &'a GpuTensorHandle, second: Option<&'a GpuTensorHandle>, weights: Option<&'a GpuTensorHandle>,
                      options: &'a CovarianceOptions) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            if second.is_some() || weights.is_some() || options.has_weight_vector || options.rows != CovRows::All {
                return Err(anyhow!("covariance: only the dense unweighted form is offloaded"));  // callers use the CPU path
            }
            let mut out = 0u64;
            let biased = matches!(options.normalization, CovNormalization::Biased) as c_int;
            check(unsafe { rmhip_covariance(self.ctx, self.own(matrix)?, biased, &mut out) })?;
            self.handle(out)
        })
    }
    fn covariance_to_correlation(&self, matrix: &GpuTensorHandle) -> Result<ProviderCovarianceToCorrelationResult> {
        let (mut corr, mut sig) = (0u64, 0u64);
        check(unsafe { rmhip_covariance_to_correlation(self.ctx, self.own(matrix)?, &mut corr, &mut sig) })?;
        Ok(ProviderCovarianceToCorrelationResult { correlation: self.handle(corr)?, sigma: self.handle(sig)? })
    }
    fn rank<'a>(&'a self, matrix: &'a GpuTensorHandle, tolerance: Option<f64>) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_rank(self.ctx, self.own(matrix)?, tolerance.is_some() as c_int, tolerance.unwrap_or(0.0), &mut out) })?;
            self.handle(out)
        })
    }
    fn cond<'a>(&'a self, matrix: &'a GpuTensorHandle, norm: ProviderCondNorm) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let code = match norm { ProviderCondNorm::Two => 0, ProviderCondNorm::One => 1, ProviderCondNorm::Inf => 2, ProviderCondNorm::Fro => 3 };
            let mut out = 0u64;
            check(unsafe { rmhip_cond(self.ctx, self.own(matrix)?, code, &mut out) })?;
            self.handle(out)
        })
    }
    fn rcond<'a>(&'a self, matrix: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_rcond(self.ctx, self.own(matrix)?, &mut out) })?;
            self.handle(out)
        })
    }
    fn pinv<'a>(&'a self, matrix: &'a GpuTensorHandle, options: ProviderPinvOptions) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_pinv(self.ctx, self.own(matrix)?, options.tolerance.is_some() as c_int, options.tolerance.unwrap_or(0.0), &mut out) })?;
            self.handle(out)
        })
    }
    fn peaks(&self, n: usize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_peaks(self.ctx, n, 0, 0, &mut out) })?;
        self.handle(out)
    }
    fn peaks_xy(&self, x: &GpuTensorHandle, y: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_peaks(self.ctx, 0, self.own(x)?, self.own(y)?, &mut out) })?;
        self.handle(out)
    }
    fn corrcoef<'a>(&'a self, matrix: &'a GpuTensorHandle, options: &'a CorrcoefOptions) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let biased = matches!(options.normalization, CorrcoefNormalization::Biased) as c_int;
            let rows = match options.rows { CorrcoefRows::All => 0, CorrcoefRows::Complete => 1, CorrcoefRows::Pairwise => 2 };
            let mut out = 0u64;
            check(unsafe { rmhip_corrcoef(self.ctx, self.own(matrix)?, biased, rows, &mut out) })?;
            self.handle(out)
        })
    }
    fn diag_extract(&self, matrix: &GpuTensorHandle, offset: isize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_diag_extract(self.ctx, self.own(matrix)?, offset as i64, &mut out) })?;
        self.handle(out)
    }
    fn syrk(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_syrk(self.ctx, self.own(a)?, &mut out) })?;
        self.handle(out)
    }
    // The library keeps the transpose lazily (a view consumed in place by matmul / syrk); nothing to record on
    // the Rust side, so `handle_transpose_info` stays empty for these handles and callers treat them as plain.
    fn transpose(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_transpose(self.ctx, self.own(a)?, &mut out) })?;
        self.handle(out)
    }
    fn lu<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, ProviderLuResult> {
        Box::pin(async move {
            let mut ids = [0u64; 5];
            check(unsafe { rmhip_lu(self.ctx, self.own(a)?, ids.as_mut_ptr()) })?;
            Ok(ProviderLuResult { combined: self.handle(ids[0])?, lower: self.handle(ids[1])?, upper: self.handle(ids[2])?,
                perm_matrix: self.handle(ids[3])?, perm_vector: self.handle(ids[4])? })
        })
    }
    fn stochastic_evolution(&self, state: &GpuTensorHandle, drift: f64, scale: f64, steps: u32) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_stochastic_evolution(self.ctx, self.own(state)?, drift, scale, steps, &mut out) })?;
        self.handle(out)
    }
    fn random_normal(&self, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_random_normal(self.ctx, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn set_rng_state(&self, state: u64) -> Result<()> { check(unsafe { rmhip_set_rng_state(self.ctx, state) }) }
    fn random_uniform(&self, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_random_uniform(self.ctx, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    // subscript / grid / slice-write hooks and the per-element forms of a real tensor (index_ops.hip)
    fn ndgrid(&self, request: &ProviderNdgridRequest<'_>) -> Result<ProviderNdgridResult> {
        let ids: Vec<u64> = request.axes.iter().map(|a| self.own(a.handle)).collect::<Result<_>>()?;
        let mut outs = vec![0u64; request.output_count.max(1)];
        check(unsafe {
            rmhip_ndgrid(self.ctx, ids.as_ptr(), ids.len(), request.output_shape.as_ptr(), request.output_shape.len(), request.output_count, outs.as_mut_ptr())
        })?;
        let outputs = outs[..request.output_count].iter().map(|&id| self.handle(id)).collect::<Result<Vec<_>>>()?;
        Ok(ProviderNdgridResult { outputs })
    }
    fn sub2ind(&self, dims: &[usize], strides: &[usize], inputs: &[&GpuTensorHandle], scalar_mask: &[bool], len: usize, output_shape: &[usize]) -> Result<GpuTensorHandle> {
        if inputs.len() != dims.len() || inputs.len() != scalar_mask.len() || strides.len() != dims.len() {
            return Err(anyhow!("sub2ind: expected {} subscripts for {} dimensions", dims.len(), dims.len()));
        }
        let ids: Vec<u64> = inputs.iter().map(|h| self.own(h)).collect::<Result<_>>()?;
        let mask: Vec<u8> = scalar_mask.iter().map(|&m| m as u8).collect();
        let mut out = 0u64;
        check(unsafe {
            rmhip_sub2ind(self.ctx, dims.as_ptr(), strides.as_ptr(), ids.as_ptr(), mask.as_ptr(), dims.len(), len, output_shape.as_ptr(), output_shape.len(), &mut out)
        })?;
        self.handle(out)
    }
    fn supports_ind2sub(&self) -> bool { true }
    fn ind2sub(&self, dims: &[usize], strides: &[usize], indices: &GpuTensorHandle, total: usize, len: usize, output_shape: &[usize]) -> Result<Vec<GpuTensorHandle>> {
        if dims.len() != strides.len() {
            return Err(anyhow!("ind2sub: size vector mismatch"));
        }
        let mut outs = vec![0u64; dims.len().max(1)];
        check(unsafe {
            rmhip_ind2sub(self.ctx, dims.as_ptr(), strides.as_ptr(), dims.len(), self.own(indices)?, total, len, output_shape.as_ptr(), output_shape.len(), outs.as_mut_ptr())
        })?;
        outs[..dims.len()].iter().map(|&id| self.handle(id)).collect()
    }
    fn scatter_column(&self, matrix: &GpuTensorHandle, col_index: usize, values: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_scatter_line(self.ctx, self.own(matrix)?, 1, col_index, self.own(values)?, &mut out) })?;
        self.handle(out)
    }
    fn scatter_row(&self, matrix: &GpuTensorHandle, row_index: usize, values: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_scatter_line(self.ctx, self.own(matrix)?, 0, row_index, self.own(values)?, &mut out) })?;
        self.handle(out)
    }
    fn pow2_scale(&self, mantissa: &GpuTensorHandle, exponent: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_pow2_scale(self.ctx, self.own(mantissa)?, self.own(exponent)?, &mut out) })?;
        self.handle(out)
    }
    fn round_digits<'a>(&'a self, a: &'a GpuTensorHandle, digits: i32, significant: bool) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_round_digits(self.ctx, self.own(a)?, digits, significant as c_int, &mut out) })?;
            self.handle(out)
        })
    }
    fn unary_real<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.real_part(0, a) }) }
    fn unary_imag<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.real_part(1, a) }) }
    fn unary_conj<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.real_part(2, a) }) }
    fn unary_angle<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.real_part(3, a) }) }
    fn logical_isreal(&self, a: &GpuTensorHandle) -> Result<bool> {
        let mut res: c_int = 0;
        check(unsafe { rmhip_isreal(self.ctx, self.own(a)?, &mut res) })?;
        Ok(res != 0)
    }
    // small construction / linear-algebra hooks (misc_ops.hip)
    fn diag_from_vector(&self, vector: &GpuTensorHandle, offset: isize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_diag_from_vector(self.ctx, self.own(vector)?, offset as i64, -1, -1, &mut out) })?;
        self.handle(out)
    }
    fn diag_from_vector_sized(&self, vector: &GpuTensorHandle, offset: isize, rows: usize, cols: usize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_diag_from_vector(self.ctx, self.own(vector)?, offset as i64, rows as i64, cols as i64, &mut out) })?;
        self.handle(out)
    }
    fn kron(&self, a: &GpuTensorHandle, b: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_kron(self.ctx, self.own(a)?, self.own(b)?, &mut out) })?;
        self.handle(out)
    }
    // `dim` arrives ONE-based (cross.rs:231-238); None -> 0 = the first dimension of extent 3
    fn cross(&self, lhs: &GpuTensorHandle, rhs: &GpuTensorHandle, dim: Option<usize>) -> Result<GpuTensorHandle> {
        if dim == Some(0) {
            return Err(anyhow!("cross: dimension must be >= 1"));
        }
        let mut out = 0u64;
        check(unsafe { rmhip_cross(self.ctx, self.own(lhs)?, self.own(rhs)?, dim.unwrap_or(0) as c_int, &mut out) })?;
        self.handle(out)
    }
    fn gradient_dim(&self, handle: &GpuTensorHandle, dim: usize, spacing: f64) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_gradient_dim(self.ctx, self.own(handle)?, dim as c_int, spacing, 0, &mut out) })?;
        self.handle(out)
    }
    fn gradient_dim_with_coordinates(&self, handle: &GpuTensorHandle, dim: usize, coordinates: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_gradient_dim(self.ctx, self.own(handle)?, dim as c_int, 1.0, self.own(coordinates)?, &mut out) })?;
        self.handle(out)
    }
    fn trapz_dim(&self, input: &GpuTensorHandle, dim: usize, spacing: ProviderTrapezoidSpacing<'_>) -> Result<GpuTensorHandle> {
        self.trapezoid(input, dim, spacing, 0)
    }
    fn cumtrapz_dim(&self, input: &GpuTensorHandle, dim: usize, spacing: ProviderTrapezoidSpacing<'_>) -> Result<GpuTensorHandle> {
        self.trapezoid(input, dim, spacing, 1)
    }
    fn norm<'a>(&'a self, tensor: &'a GpuTensorHandle, order: ProviderNormOrder) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let (code, p) = match order {
                ProviderNormOrder::One => (1, 0.0),
                ProviderNormOrder::Two => (2, 0.0),
                ProviderNormOrder::Inf => (3, 0.0),
                ProviderNormOrder::NegInf => (4, 0.0),
                ProviderNormOrder::Zero => (5, 0.0),
                ProviderNormOrder::Fro => (6, 0.0),
                ProviderNormOrder::Nuc => (7, 0.0),
                ProviderNormOrder::P(p) => (8, p),
            };
            let mut out = 0u64;
            check(unsafe { rmhip_norm(self.ctx, self.own(tensor)?, code, p, &mut out) })?;
            self.handle(out)
        })
    }
    fn issymmetric(&self, matrix: &GpuTensorHandle, kind: ProviderSymmetryKind, tolerance: f64) -> Result<bool> {
        let mut res: c_int = 0;
        let skew = matches!(kind, ProviderSymmetryKind::Skew) as c_int;
        check(unsafe { rmhip_issymmetric(self.ctx, self.own(matrix)?, skew, tolerance, &mut res) })?;
        Ok(res != 0)
    }
    fn unique<'a>(&'a self, handle: &'a GpuTensorHandle, options: &'a UniqueOptions) -> AccelProviderFuture<'a, UniqueResult> {
        Box::pin(async move {
            if options.rows {
                return Err(anyhow!("unique: the 'rows' form is not served by this provider"));
            }
            let n: usize = handle.shape.iter().product();
            let (mut values, mut ia, mut ic) = (vec![0.0f64; n], vec![0.0f64; n], vec![0.0f64; n]);
            let mut count = 0usize;
            let stable = matches!(options.order, UniqueOrder::Stable) as c_int;
            let last = matches!(options.occurrence, UniqueOccurrence::Last) as c_int;
            check(unsafe { rmhip_unique(self.ctx, self.own(handle)?, stable, last, &mut count, values.as_mut_ptr(), ia.as_mut_ptr(), ic.as_mut_ptr()) })?;
            values.truncate(count);
            ia.truncate(count);
            let real = GpuTensorStorage::Real;
            Ok(UniqueResult {
                values: HostTensorOwned { data: values, shape: vec![count, 1], storage: real },
                ia: HostTensorOwned { data: ia, shape: vec![count, 1], storage: real },
                ic: HostTensorOwned { data: ic, shape: vec![n, 1], storage: real },
            })
        })
    }
    fn union<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle, options: &'a UnionOptions) -> AccelProviderFuture<'a, UnionResult> {
        Box::pin(async move {
            if options.rows {
                return Err(anyhow!("union: the 'rows' form is not served by this provider"));
            }
            let (na, nb): (usize, usize) = (a.shape.iter().product(), b.shape.iter().product());
            let (mut values, mut ia, mut ib) = (vec![0.0f64; na + nb], vec![0.0f64; na], vec![0.0f64; nb]);
            let (mut n, mut ca, mut cb) = (0usize, 0usize, 0usize);
            let stable = matches!(options.order, UnionOrder::Stable) as c_int;
            check(unsafe {
                rmhip_union(self.ctx, self.own(a)?, self.own(b)?, stable, &mut n, values.as_mut_ptr(), &mut ca, ia.as_mut_ptr(), &mut cb, ib.as_mut_ptr())
            })?;
            values.truncate(n);
            ia.truncate(ca);
            ib.truncate(cb);
            let real = GpuTensorStorage::Real;
            Ok(UnionResult {
                values: HostTensorOwned { data: values, shape: vec![n, 1], storage: real },
                ia: HostTensorOwned { data: ia, shape: vec![ca, 1], storage: real },
                ib: HostTensorOwned { data: ib, shape: vec![cb, 1], storage: real },
            })
        })
    }
    fn setdiff<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle, options: &'a SetdiffOptions) -> AccelProviderFuture<'a, SetdiffResult> {
        Box::pin(async move {
            if options.rows {
                return Err(anyhow!("setdiff: the 'rows' form is not served by this provider"));
            }
            let na: usize = a.shape.iter().product();
            let (mut values, mut ia) = (vec![0.0f64; na], vec![0.0f64; na]);
            let mut n = 0usize;
            let stable = matches!(options.order, SetdiffOrder::Stable) as c_int;
            check(unsafe { rmhip_setdiff(self.ctx, self.own(a)?, self.own(b)?, stable, &mut n, values.as_mut_ptr(), ia.as_mut_ptr()) })?;
            values.truncate(n);
            ia.truncate(n);
            let real = GpuTensorStorage::Real;
            Ok(SetdiffResult {
                values: HostTensorOwned { data: values, shape: vec![n, 1], storage: real },
                ia: HostTensorOwned { data: ia, shape: vec![n, 1], storage: real },
            })
        })
    }
    fn ismember<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle, options: &'a IsMemberOptions) -> AccelProviderFuture<'a, IsMemberResult> {
        Box::pin(async move {
            if options.rows {
                return Err(anyhow!("ismember: the 'rows' form is not served by this provider"));
            }
            let n: usize = a.shape.iter().product();
            let (mut mask, mut loc) = (vec![0u8; n], vec![0.0f64; n]);
            check(unsafe { rmhip_ismember(self.ctx, self.own(a)?, self.own(b)?, mask.as_mut_ptr(), loc.as_mut_ptr()) })?;
            Ok(IsMemberResult {
                mask: HostLogicalOwned { data: mask, shape: a.shape.clone() },
                loc: HostTensorOwned { data: loc, shape: a.shape.clone(), storage: GpuTensorStorage::Real },
            })
        })
    }
    fn imfilter<'a>(&'a self, image: &'a GpuTensorHandle, kernel: &'a GpuTensorHandle, options: &'a ImfilterOptions) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let padding = match options.padding { ImfilterPadding::Constant => 0, ImfilterPadding::Replicate => 1, ImfilterPadding::Symmetric => 2, ImfilterPadding::Circular => 3 };
            let shape = match options.shape { ImfilterShape::Same => 0, ImfilterShape::Full => 1, ImfilterShape::Valid => 2 };
            let conv = matches!(options.mode, ImfilterMode::Convolution) as c_int;
            let mut out = 0u64;
            check(unsafe { rmhip_imfilter(self.ctx, self.own(image)?, self.own(kernel)?, padding, options.constant_value, shape, conv, &mut out) })?;
            self.handle(out)
        })
    }
    fn interp1<'a>(&'a self, request: &'a ProviderInterp1Request<'a>) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let nearest = matches!(request.method, ProviderInterp1Method::Nearest) as c_int;
            let mode = match request.extrapolation {
                ProviderInterp1Extrapolation::Nan => 0,
                ProviderInterp1Extrapolation::Extrapolate => 1,
                ProviderInterp1Extrapolation::Value => 2,
            };
            let mut out = 0u64;
            check(unsafe {
                rmhip_interp1(self.ctx, self.own(request.x)?, self.own(request.y)?, self.own(request.xq)?, request.sample_len, request.series_count, request.query_len,
                              request.output_shape.as_ptr(), request.output_shape.len(), nearest, mode, request.extrapolation_value, &mut out)
            })?;
            self.handle(out)
        })
    }
    fn iir_filter<'a>(&'a self, b: &'a GpuTensorHandle, a: &'a GpuTensorHandle, x: &'a GpuTensorHandle, options: ProviderIirFilterOptions)
        -> AccelProviderFuture<'a, ProviderIirFilterResult> {
        Box::pin(async move {
            let (mut out, mut fin) = (0u64, 0u64);
            let zi = match &options.zi { Some(h) => self.own(h)?, None => 0 };
            check(unsafe {
                rmhip_iir_filter(self.ctx, self.own(b)?, self.own(a)?, self.own(x)?, options.dim as c_int, zi, options.unit_denominator as c_int, &mut out, &mut fin)
            })?;
            Ok(ProviderIirFilterResult { output: self.handle(out)?, final_state: Some(self.handle(fin)?) })
        })
    }
    fn polyval(&self, coefficients: &GpuTensorHandle, points: &GpuTensorHandle, options: &ProviderPolyvalOptions) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        let (has_mu, mean, scale) = match options.mu { Some(mu) => (1, mu.mean, mu.scale), None => (0, 0.0, 1.0) };
        check(unsafe { rmhip_polyval(self.ctx, self.own(coefficients)?, self.own(points)?, has_mu, mean, scale, &mut out) })?;
        self.handle(out)
    }
    fn polyder_single<'a>(&'a self, polynomial: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_polyder(self.ctx, self.own(polynomial)?, 0, 0, &mut out, std::ptr::null_mut()) })?;
            self.handle(out)
        })
    }
    fn polyder_product<'a>(&'a self, p: &'a GpuTensorHandle, q: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_polyder(self.ctx, self.own(p)?, self.own(q)?, 0, &mut out, std::ptr::null_mut()) })?;
            self.handle(out)
        })
    }
    fn polyder_quotient<'a>(&'a self, u: &'a GpuTensorHandle, v: &'a GpuTensorHandle) -> AccelProviderFuture<'a, ProviderPolyderQuotient> {
        Box::pin(async move {
            let (mut num, mut den) = (0u64, 0u64);
            check(unsafe { rmhip_polyder(self.ctx, self.own(u)?, self.own(v)?, 1, &mut num, &mut den) })?;
            Ok(ProviderPolyderQuotient { numerator: self.handle(num)?, denominator: self.handle(den)? })
        })
    }
    fn polyint(&self, polynomial: &GpuTensorHandle, constant: f64) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_polyint(self.ctx, self.own(polynomial)?, constant, &mut out) })?;
        self.handle(out)
    }
    fn meshgrid(&self, axes: &[MeshgridAxisView<'_>]) -> Result<ProviderMeshgridResult> {
        if axes.len() != 2 && axes.len() != 3 {
            return Err(anyhow!("meshgrid: provider expects two or three axes"));
        }
        let mut outs = [0u64; 3];
        let (zp, zn) = match axes.get(2) { Some(z) => (z.data.as_ptr(), z.data.len()), None => (std::ptr::null(), 0) };
        check(unsafe { rmhip_meshgrid(self.ctx, axes[0].data.as_ptr(), axes[0].data.len(), axes[1].data.as_ptr(), axes[1].data.len(), zp, zn, outs.as_mut_ptr()) })?;
        let outputs = outs[..axes.len()].iter().map(|&id| self.handle(id)).collect::<Result<Vec<_>>>()?;
        Ok(ProviderMeshgridResult { outputs })
    }
    fn zeros_with_storage(&self, shape: &[usize], storage: GpuTensorStorage) -> Result<GpuTensorHandle> {
        if storage == GpuTensorStorage::Real {
            return self.zeros(shape);
        }
        let mut out = 0u64;
        check(unsafe { rmhip_zeros_complex(self.ctx, shape.as_ptr(), shape.len(), &mut out) })?;
        self.complex_handle(out)
    }
    fn moving_window<'a>(&'a self, request: &'a ProviderMovingWindowRequest<'a>) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let op = match request.op {
                ProviderMovingWindowOp::Sum => 0, ProviderMovingWindowOp::Mean => 1, ProviderMovingWindowOp::Prod => 2, ProviderMovingWindowOp::Min => 3,
                ProviderMovingWindowOp::Max => 4, ProviderMovingWindowOp::Median => 5, ProviderMovingWindowOp::Std => 6, ProviderMovingWindowOp::Var => 7,
            };
            let (endpoints, fill) = match request.endpoints {
                ProviderMovingWindowEndpoints::Shrink => (0, 0.0),
                ProviderMovingWindowEndpoints::Discard => (1, 0.0),
                ProviderMovingWindowEndpoints::Fill(v) => (2, v),
            };
            let omit = matches!(request.nan_mode, ProviderNanMode::Omit) as c_int;
            let population = matches!(request.normalization, ProviderStdNormalization::Population) as c_int;
            let mut out = 0u64;
            check(unsafe {
                rmhip_moving_window(self.ctx, self.own(request.input)?, request.dim as c_int, request.before, request.after, op, endpoints, fill, omit, population,
                                    request.output_shape.as_ptr(), request.output_shape.len(), &mut out)
            })?;
            self.handle(out)
        })
    }
    fn conv1d(&self, signal: &GpuTensorHandle, kernel: &GpuTensorHandle, options: ProviderConv1dOptions) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        let column = matches!(options.orientation, ProviderConvOrientation::Column) as c_int;
        check(unsafe { rmhip_conv1d(self.ctx, self.own(signal)?, self.own(kernel)?, conv_mode(options.mode), column, &mut out) })?;
        self.handle(out)
    }
    fn conv2d(&self, signal: &GpuTensorHandle, kernel: &GpuTensorHandle, mode: ProviderConvMode) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_conv2d(self.ctx, self.own(signal)?, self.own(kernel)?, conv_mode(mode), &mut out) })?;
        self.handle(out)
    }
    fn hann_window(&self, len: usize, periodic: bool) -> Result<GpuTensorHandle> { self.window(0, len, periodic) }
    fn hamming_window(&self, len: usize, periodic: bool) -> Result<GpuTensorHandle> { self.window(1, len, periodic) }
    fn blackman_window(&self, len: usize, periodic: bool) -> Result<GpuTensorHandle> { self.window(2, len, periodic) }
    // transforms -> complex-interleaved tensors; the storage kind is recorded for the callers that ask `handle_storage` (lib.rs:582-594)
    fn fft_dim<'a>(&'a self, handle: &'a GpuTensorHandle, len: Option<usize>, dim: usize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.transform(handle, len, dim, 0) })
    }
    fn ifft_dim<'a>(&'a self, handle: &'a GpuTensorHandle, len: Option<usize>, dim: usize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.transform(handle, len, dim, 1) })
    }
    fn signal_hilbert<'a>(&'a self, request: &'a ProviderHilbertRequest<'a>) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            let len = request.length.map(|l| l as i64).unwrap_or(-1);
            check(unsafe { rmhip_hilbert(self.ctx, self.own(request.input)?, len, request.dim as c_int, &mut out) })?;
            self.complex_handle(out)
        })
    }
    fn fft_extract_real<'a>(&'a self, handle: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_complex_real(self.ctx, self.own(handle)?, &mut out) })?;
            self.handle(out)
        })
    }
    fn complex_from_real<'a>(&'a self, real: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.complex(real, None) })
    }
    fn complex_from_real_imag<'a>(&'a self, real: &'a GpuTensorHandle, imag: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.complex(real, Some(imag)) })
    }
    fn ishermitian<'a>(&'a self, matrix: &'a GpuTensorHandle, kind: ProviderHermitianKind, tolerance: f64) -> AccelProviderFuture<'a, bool> {
        Box::pin(async move {
            let mut res: c_int = 0;
            let skew = matches!(kind, ProviderHermitianKind::Skew) as c_int;
            check(unsafe { rmhip_ishermitian(self.ctx, self.own(matrix)?, skew, tolerance, &mut res) })?;
            Ok(res != 0)
        })
    }
    fn bandwidth(&self, matrix: &GpuTensorHandle) -> Result<ProviderBandwidth> {
        let (mut lower, mut upper) = (0u32, 0u32);
        check(unsafe { rmhip_bandwidth(self.ctx, self.own(matrix)?, &mut lower, &mut upper) })?;
        Ok(ProviderBandwidth { lower, upper })
    }
    // the prototype forms (lib.rs:1718-1730, 1832-1839: the trait's defaults, spelled out) and the scaled / transformed draws
    fn random_uniform_like(&self, prototype: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.random_uniform(&prototype.shape) }
    fn random_normal_like(&self, prototype: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.random_normal(&prototype.shape) }
    fn random_unifrnd(&self, a: f64, b: f64, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_random_unifrnd(self.ctx, a, b, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn random_exponential(&self, mu: f64, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_random_exponential(self.ctx, mu, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn random_normrnd(&self, mu: f64, sigma: f64, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_random_normrnd(self.ctx, mu, sigma, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn random_integer_range(&self, lower: i64, upper: i64, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_random_integer_range(self.ctx, lower, upper, shape.as_ptr(), shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn random_integer_like(&self, prototype: &GpuTensorHandle, lower: i64, upper: i64) -> Result<GpuTensorHandle> {
        self.random_integer_range(lower, upper, &prototype.shape)
    }

    // all-element truth reductions (lib.rs:2730-2735, 2803-2809, 2818-2824): dim = -1
    fn reduce_nnz<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.truth(RMHIP_TNNZ, a, -1, false) })
    }
    fn reduce_any<'a>(&'a self, a: &'a GpuTensorHandle, omit_nan: bool) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.truth(RMHIP_TANY, a, -1, omit_nan) })
    }
    fn reduce_all<'a>(&'a self, a: &'a GpuTensorHandle, omit_nan: bool) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move { self.truth(RMHIP_TALL, a, -1, omit_nan) })
    }

    // `matmul_epilogue` (lib.rs:2394-2405): the descriptor's Options become flags, absent handles buffer id 0
    fn matmul_epilogue<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle, epilogue: &'a MatmulEpilogue)
        -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let id_of = |h: &Option<GpuTensorHandle>| -> Result<u64> { h.as_ref().map(|h| self.own(h)).transpose().map(|v| v.unwrap_or(0)) };
            let ep = RmhipMatmulEpilogue {
                alpha: epilogue.alpha,
                beta: epilogue.beta,
                row_scale: id_of(&epilogue.row_scale)?,
                col_scale: id_of(&epilogue.col_scale)?,
                row_op: matches!(epilogue.row_op, ScaleOp::Divide) as c_int,
                col_op: matches!(epilogue.col_op, ScaleOp::Divide) as c_int,
                has_clamp_min: epilogue.clamp_min.is_some() as c_int,
                has_clamp_max: epilogue.clamp_max.is_some() as c_int,
                has_pow: epilogue.pow_exponent.is_some() as c_int,
                clamp_min: epilogue.clamp_min.unwrap_or(0.0),
                clamp_max: epilogue.clamp_max.unwrap_or(0.0),
                pow_exponent: epilogue.pow_exponent.unwrap_or(1.0),
                diag_output: id_of(&epilogue.diag_output)?,
            };
            let mut out = 0u64;
            check(unsafe { rmhip_matmul_epilogue(self.ctx, self.own(a)?, self.own(b)?, &ep, &mut out) })?;
            self.handle(out)
        })
    }

    // shape / indexing hooks the hot-path builtins call around the kernels (include/rmhip.h "shape / indexing hooks")
    fn repmat(&self, handle: &GpuTensorHandle, reps: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64; // a view: elem_* / fused_elementwise read it in place (times.rs:501-543)
        check(unsafe { rmhip_repmat(self.ctx, self.own(handle)?, reps.as_ptr(), reps.len(), &mut out) })?;
        self.handle(out)
    }
    fn permute(&self, handle: &GpuTensorHandle, order: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_permute(self.ctx, self.own(handle)?, order.as_ptr(), order.len(), &mut out) })?;
        self.handle(out)
    }
    fn fill_like(&self, prototype: &GpuTensorHandle, value: f64) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_fill_like(self.ctx, self.own(prototype)?, value, &mut out) })?;
        Ok(GpuTensorHandle { shape: prototype.shape.clone(), device_id: self.device_id, buffer_id: out })
    }
    fn zeros_like(&self, prototype: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.fill_like(prototype, 0.0) }
    fn ones_like(&self, prototype: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.fill_like(prototype, 1.0) }
    fn read_scalar(&self, h: &GpuTensorHandle, linear_index: usize) -> Result<f64> {
        let mut v = 0.0f64;
        check(unsafe { rmhip_read_scalar(self.ctx, self.own(h)?, linear_index, &mut v) })?;
        Ok(v)
    }
    fn gather_linear(&self, source: &GpuTensorHandle, indices: &[u32], output_shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_gather_linear(self.ctx, self.own(source)?, indices.as_ptr(), indices.len(), output_shape.as_ptr(),
            output_shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: output_shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn scatter_linear(&self, target: &GpuTensorHandle, indices: &[u32], values: &GpuTensorHandle) -> Result<()> {
        check(unsafe { rmhip_scatter_linear(self.ctx, self.own(target)?, indices.as_ptr(), indices.len(), self.own(values)?) })
    }
    fn linspace(&self, start: f64, stop: f64, count: usize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_linspace(self.ctx, start, stop, count, &mut out) })?;
        Ok(GpuTensorHandle { shape: vec![1, count], device_id: self.device_id, buffer_id: out })
    }
    fn eye(&self, shape: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_eye(self.ctx, shape.as_ptr(), shape.len(), &mut out) })?;
        self.handle(out)
    }
    fn eye_like(&self, prototype: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.eye(&prototype.shape) }
    fn flip(&self, handle: &GpuTensorHandle, axes: &[usize]) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_flip(self.ctx, self.own(handle)?, axes.as_ptr(), axes.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: handle.shape.clone(), device_id: self.device_id, buffer_id: out })
    }
    fn circshift(&self, handle: &GpuTensorHandle, shifts: &[isize]) -> Result<GpuTensorHandle> {
        let s: Vec<std::ffi::c_longlong> = shifts.iter().map(|&v| v as std::ffi::c_longlong).collect();
        let mut out = 0u64;
        check(unsafe { rmhip_circshift(self.ctx, self.own(handle)?, s.as_ptr(), s.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: handle.shape.clone(), device_id: self.device_id, buffer_id: out })
    }
    fn tril<'a>(&'a self, matrix: &'a GpuTensorHandle, offset: isize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_tri(self.ctx, self.own(matrix)?, 0, offset as std::ffi::c_longlong, &mut out) })?;
            Ok(GpuTensorHandle { shape: matrix.shape.clone(), device_id: self.device_id, buffer_id: out })
        })
    }
    fn triu<'a>(&'a self, matrix: &'a GpuTensorHandle, offset: isize) -> AccelProviderFuture<'a, GpuTensorHandle> {
        Box::pin(async move {
            let mut out = 0u64;
            check(unsafe { rmhip_tri(self.ctx, self.own(matrix)?, 1, offset as std::ffi::c_longlong, &mut out) })?;
            Ok(GpuTensorHandle { shape: matrix.shape.clone(), device_id: self.device_id, buffer_id: out })
        })
    }
    fn cat(&self, dim: usize, inputs: &[GpuTensorHandle]) -> Result<GpuTensorHandle> {
        let ids = inputs.iter().map(|h| self.own(h)).collect::<Result<Vec<_>>>()?;
        let mut out = 0u64;
        check(unsafe { rmhip_cat(self.ctx, dim, ids.as_ptr(), ids.len(), &mut out) })?;
        self.handle(out)
    }
    fn map_nan_to_zero(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.unary(RMHIP_NAN_TO_ZERO, a) }
    fn not_nan_mask(&self, a: &GpuTensorHandle) -> Result<GpuTensorHandle> { self.unary(RMHIP_NOT_NAN, a) }

    // identity / telemetry (lib.rs:1390, 1448-1456, 3014-3056)
    fn device_info(&self) -> String {
        let i = self.raw_device_info();
        format!("{} ({}, {})", cstr(&i.name), cstr(&i.arch), cstr(&i.backend))
    }
    fn device_info_struct(&self) -> ApiDeviceInfo {
        let i = self.raw_device_info();
        ApiDeviceInfo { device_id: self.device_id, name: cstr(&i.name), vendor: cstr(&i.vendor), memory_bytes: Some(i.total_memory_bytes),
            backend: Some(cstr(&i.backend)) }
    }
    fn default_reduction_workgroup_size(&self) -> u32 { self.raw_device_info().reduction_workgroup_size }
    fn two_pass_threshold(&self) -> usize { self.raw_device_info().two_pass_threshold as usize }
    fn fused_cache_counters(&self) -> (u64, u64) {
        let t = self.raw_telemetry();
        (t.fusion_cache_hits, t.fusion_cache_misses)
    }
    fn telemetry_snapshot(&self) -> ProviderTelemetry {
        let t = self.raw_telemetry();
        let stats = |count: u64, ns: u64| ProviderDispatchStats { count, total_wall_time_ns: ns };
        // solve_fallbacks: (reason, count) pairs until the index runs out (RMHIP_ERR_NOT_FOUND)
        let mut solve_fallbacks = Vec::new();
        let mut reason = [0 as c_char; 64];
        let mut count = 0u64;
        let mut index = 0usize;
        while unsafe { rmhip_telemetry_solve_fallback(self.ctx, index, reason.as_mut_ptr(), reason.len(), &mut count) } == 0 {
            solve_fallbacks.push(ProviderFallbackStat { reason: cstr(&reason), count });
            index += 1;
        }
        // kernel_launches: bounded log, oldest first
        let mut kernel_launches = Vec::new();
        let mut rec: RmhipKernelLaunch = unsafe { std::mem::zeroed() };
        let mut index = 0usize;
        while unsafe { rmhip_telemetry_kernel_launch(self.ctx, index, &mut rec) } == 0 {
            let attrs = |a: &[RmhipKernelAttr; 6], n: u32| a[..n as usize].iter()
                .map(|kv| KernelAttrTelemetry { key: cstr(&kv.key), value: kv.value }).collect::<Vec<_>>();
            kernel_launches.push(KernelLaunchTelemetry { kernel: cstr(&rec.kernel), precision: Some(cstr(&rec.precision)),
                shape: attrs(&rec.shape, rec.n_shape), tuning: attrs(&rec.tuning, rec.n_tuning) });
            index += 1;
        }
        ProviderTelemetry {
            fused_elementwise: stats(t.fused_elementwise_count, t.fused_elementwise_ns),
            fused_reduction: stats(t.fused_reduction_count, t.fused_reduction_ns),
            matmul: stats(t.matmul_count, t.matmul_ns),
            linsolve: stats(t.linsolve_count, t.linsolve_ns),
            mldivide: stats(t.mldivide_count, t.mldivide_ns),
            mrdivide: stats(t.mrdivide_count, t.mrdivide_ns),
            upload_bytes: t.upload_bytes,
            download_bytes: t.download_bytes,
            solve_fallbacks,
            fusion_cache_hits: t.fusion_cache_hits,
            fusion_cache_misses: t.fusion_cache_misses,
            bind_group_cache_hits: 0, // no bind groups on this backend
            bind_group_cache_misses: 0,
            bind_group_cache_by_layout: None,
            kernel_launches,
        }
    }
    fn reset_telemetry(&self) { unsafe { rmhip_reset_telemetry(self.ctx) }; }
}

fn cstr(bytes: &[c_char]) -> String {
    let end = bytes.iter().position(|&b| b == 0).unwrap_or(bytes.len());
    String::from_utf8_lossy(&bytes[..end].iter().map(|&b| b as u8).collect::<Vec<u8>>()).into_owned()
}

impl HipProvider {
    fn raw_device_info(&self) -> RmhipDeviceInfo {
        let mut info: RmhipDeviceInfo = unsafe { std::mem::zeroed() };
        unsafe { rmhip_device_info(self.ctx, &mut info) };
        info
    }
    fn raw_telemetry(&self) -> RmhipTelemetry {
        let mut t: RmhipTelemetry = unsafe { std::mem::zeroed() };
        unsafe { rmhip_telemetry(self.ctx, &mut t) };
        t
    }
    fn truth(&self, op: c_int, a: &GpuTensorHandle, dim: c_int, omit_nan: bool) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_reduce_truth(self.ctx, op, self.own(a)?, dim, omit_nan as c_int, &mut out) })?;
        self.handle(out)
    }
}

/// Sharded forms of the hot path: one process per GPU, one provider per process.  The trait has no
/// multi-device surface (the reference has none), so these are inherent methods a multi-GPU host calls
/// around the trait calls: rank 0 makes an id, the host ships its 128 bytes to every rank (MPI, a file, ...), every
/// rank calls `comm_init`; then e.g. `matmul` on the local row block followed by `comm_allgather_rows`.
impl HipProvider {
    pub fn comm_unique_id(rccl: bool) -> Result<[u8; RMHIP_COMM_ID_BYTES]> {
        let mut id = [0u8; RMHIP_COMM_ID_BYTES];
        check(unsafe { rmhip_comm_unique_id(if rccl { RMHIP_COMM_RCCL } else { RMHIP_COMM_HOST_SHM }, id.as_mut_ptr() as *mut c_void) })?;
        Ok(id)
    }
    pub fn comm_init(&self, id: &[u8; RMHIP_COMM_ID_BYTES], rank: i32, world: i32) -> Result<()> {
        check(unsafe { rmhip_comm_init(self.ctx, id.as_ptr() as *const c_void, rank, world) })
    }
    pub fn comm_destroy(&self) -> Result<()> { check(unsafe { rmhip_comm_destroy(self.ctx) }) }
    pub fn comm_rank(&self) -> Result<(i32, i32)> {
        let (mut r, mut w) = (0, 1);
        check(unsafe { rmhip_comm_rank(self.ctx, &mut r, &mut w) })?;
        Ok((r, w))
    }
    pub fn comm_barrier(&self) -> Result<()> { check(unsafe { rmhip_comm_barrier(self.ctx) }) }
    /// Leave a sequence of collectives without blocking the peers (their next barrier fails); the communicator is unusable afterwards.
    pub fn comm_abort(&self) -> Result<()> { check(unsafe { rmhip_comm_abort(self.ctx) }) }
    /// In-place broadcast of a sub-block from `root`; `asynchronous` posts it on the communication stream (the
    /// block-cyclic solver's look-ahead) and `comm_wait` joins it before the block is read.
    pub fn comm_bcast(&self, block: &RmhipView, root: i32, asynchronous: bool) -> Result<()> {
        check(unsafe { rmhip_comm_bcast(self.ctx, block, root, asynchronous as c_int) })
    }
    pub fn comm_wait(&self) -> Result<()> { check(unsafe { rmhip_comm_wait(self.ctx) }) }
    /// [k] per rank -> [k, world] on every rank, column r = rank r: the caller sums the columns in rank order.
    pub fn comm_allgather_f64(&self, local: &GpuTensorHandle) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_comm_allgather_f64(self.ctx, self.own(local)?, &mut out) })?;
        self.handle(out)
    }
    /// Row blocks (balanced split of `rows_total` in units of `granule`) -> the replicated matrix.
    pub fn comm_allgather_rows(&self, local: &GpuTensorHandle, rows_total: usize, granule: usize) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_comm_allgather_rows(self.ctx, self.own(local)?, rows_total, granule, &mut out) })?;
        self.handle(out)
    }
}

/// Register before `initialize_acceleration_provider_with` so A/lib.rs:179-181 short-circuits.
pub fn register_hip_provider(device_ordinal: i32) -> Result<()> {
    let provider: &'static HipProvider = Box::leak(Box::new(HipProvider::new(device_ordinal)?));
    unsafe { runmat_accelerate_api::register_provider(provider) }; // lib.rs:3213
    Ok(())
}
