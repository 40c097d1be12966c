//! hip_provider.rs -- the binding a RunMat maintainer adds to plug librmhip.so in as a third
//! `AccelProvider` backend (next to `WgpuProvider` and `InProcessProvider`).
//!
//! NOT compiled in this repository (the build image has no Rust toolchain); it documents, in the
//! reference's own language, the exact FFI surface of `include/rmhip.h`.  Drop it into
//! `crates/runmat-accelerate/src/backend/hip/mod.rs`, add `links = "rmhip"` / a build.rs that
//! emits `cargo:rustc-link-lib=dylib=rmhip`, and call `register_hip_provider()` before
//! `initialize_acceleration_provider_with` (which returns early when a provider is already
//! registered, crates/runmat-accelerate/src/lib.rs:179-181).
//!
//! Every method maps 1:1 onto one C entry point; any non-zero status becomes `Err(anyhow!(..))`,
//! which RunMat's callers already treat as "fall back to the CPU builtin"
//! (mtimes.rs:212-216, mldivide.rs:223-229, runner.rs:1140-1142).

use anyhow::{anyhow, Result};
use runmat_accelerate_api::{
    AccelProvider, AccelProviderFuture, GpuTensorHandle, HostTensorOwned, HostTensorView,
    ProviderLuResult, ProviderMoments2, ProviderPrecision, ReductionFlavor,
};
use std::ffi::{c_char, c_double, c_int, c_void, CStr, CString};

#[repr(C)]
pub struct RmhipCtx {
    _private: [u8; 0],
}

extern "C" {
    fn rmhip_last_error() -> *const c_char;
    fn rmhip_init(device_ordinal: c_int, out: *mut *mut RmhipCtx) -> c_int;
    fn rmhip_shutdown(ctx: *mut RmhipCtx) -> c_int;
    fn rmhip_set_precision(ctx: *mut RmhipCtx, bits: c_int) -> c_int;
    fn rmhip_upload(ctx: *mut RmhipCtx, host: *const c_double, shape: *const usize, rank: usize, out: *mut u64) -> c_int;
    fn rmhip_download(ctx: *mut RmhipCtx, id: u64, out: *mut c_double, n: usize) -> c_int;
    fn rmhip_free(ctx: *mut RmhipCtx, id: u64) -> c_int;
    fn rmhip_shape(ctx: *mut RmhipCtx, id: u64, rank_inout: *mut usize, shape_out: *mut usize) -> c_int;
    fn rmhip_fill(ctx: *mut RmhipCtx, value: c_double, shape: *const usize, rank: usize, out: *mut u64) -> c_int;
    fn rmhip_fused_elementwise(ctx: *mut RmhipCtx, shader: *const c_char, inputs: *const u64, n_in: usize,
        out_shape: *const usize, rank: usize, len: usize, n_out: usize, out_ids: *mut u64) -> c_int;
    fn rmhip_fused_reduction(ctx: *mut RmhipCtx, shader: *const c_char, inputs: *const u64, n_in: usize,
        out_shape: *const usize, rank: usize, reduce_len: usize, num_slices: usize, workgroup_size: u32,
        flavor: c_int, custom_scale: c_double, out: *mut u64) -> c_int;
    fn rmhip_binary(ctx: *mut RmhipCtx, op: c_int, a: u64, b: u64, out: *mut u64) -> c_int;
    fn rmhip_unary(ctx: *mut RmhipCtx, op: c_int, a: u64, out: *mut u64) -> c_int;
    fn rmhip_scalar(ctx: *mut RmhipCtx, op: c_int, a: u64, s: c_double, out: *mut u64) -> c_int;
    fn rmhip_reduce(ctx: *mut RmhipCtx, op: c_int, a: u64, dim: c_int, nan_mode: c_int, out: *mut u64) -> c_int;
    fn rmhip_reduce_nd(ctx: *mut RmhipCtx, op: c_int, a: u64, dims: *const usize, ndims: usize, nan_mode: c_int, out: *mut u64) -> c_int;
    fn rmhip_reduce_moments_nd(ctx: *mut RmhipCtx, a: u64, dims: *const usize, ndims: usize, mean: *mut u64, ex2: *mut u64) -> c_int;
    fn rmhip_matmul(ctx: *mut RmhipCtx, a: u64, b: u64, out: *mut u64) -> c_int;
    fn rmhip_lu(ctx: *mut RmhipCtx, a: u64, out5: *mut u64) -> c_int;
    fn rmhip_mldivide(ctx: *mut RmhipCtx, a: u64, b: u64, out: *mut u64) -> c_int;
    fn rmhip_linsolve(ctx: *mut RmhipCtx, a: u64, b: u64, opts: *const RmhipLinsolveOptions, out: *mut u64, rcond: *mut c_double) -> c_int;
    fn rmhip_transpose(ctx: *mut RmhipCtx, a: u64, out: *mut u64) -> c_int;
    fn rmhip_syrk(ctx: *mut RmhipCtx, a: u64, out: *mut u64) -> c_int;
    fn rmhip_covariance(ctx: *mut RmhipCtx, matrix: u64, biased: c_int, out: *mut u64) -> c_int;
    fn rmhip_diag_extract(ctx: *mut RmhipCtx, matrix: u64, offset: i64, out: *mut u64) -> c_int;
    fn rmhip_matmul_power_step(ctx: *mut RmhipCtx, lhs: u64, rhs: u64, epsilon: c_double, out: *mut u64) -> c_int;
    fn rmhip_image_normalize(ctx: *mut RmhipCtx, input: u64, desc: *const RmhipImageNormalize, out: *mut u64) -> c_int;
    fn rmhip_set_rng_state(ctx: *mut RmhipCtx, state: u64) -> c_int;
    fn rmhip_stochastic_evolution(ctx: *mut RmhipCtx, state: u64, drift: c_double, scale: c_double, steps: u32, out: *mut u64) -> c_int;
    fn rmhip_random_normal(ctx: *mut RmhipCtx, shape: *const usize, rank: usize, out: *mut u64) -> c_int;
    fn rmhip_random_uniform(ctx: *mut RmhipCtx, shape: *const usize, rank: usize, out: *mut u64) -> c_int;
}

#[repr(C)]
struct RmhipImageNormalize {
    batch: usize, height: usize, width: usize, epsilon: c_double,
    has_gain: c_int, has_bias: c_int, has_gamma: c_int, clamp_zero: c_int,
    gain: c_double, bias: c_double, gamma: c_double,
}

#[repr(C)]
struct RmhipLinsolveOptions {
    lower: c_int, upper: c_int, rectangular: c_int, transposed: c_int, conjugate: c_int, symmetric: c_int, posdef: c_int,
    need_rcond: c_int, has_rcond: c_int, rcond: c_double,
}

pub struct HipProvider {
    ctx: *mut RmhipCtx,
    precision: ProviderPrecision,
    device_id: u32,
}
// One HIP stream per context; the library serialises table access internally.
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

impl AccelProvider for HipProvider {
    fn upload(&self, host: &HostTensorView) -> Result<GpuTensorHandle> {
        let mut out = 0u64;
        check(unsafe { rmhip_upload(self.ctx, host.data.as_ptr(), host.shape.as_ptr(), host.shape.len(), &mut out) })?;
        Ok(GpuTensorHandle { shape: host.shape.to_vec(), device_id: self.device_id, buffer_id: out })
    }
    fn download<'a>(&'a self, h: &'a GpuTensorHandle) -> AccelProviderFuture<'a, HostTensorOwned> {
        Box::pin(async move {
            let n: usize = h.shape.iter().product();
            let mut data = vec![0.0f64; n];
            check(unsafe { rmhip_download(self.ctx, self.own(h)?, data.as_mut_ptr(), n) })?;
            Ok(HostTensorOwned { data, shape: h.shape.clone() })
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

    // per-op hooks: op codes are the enums of include/rmhip.h
    fn elem_add<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.binary(0, a, b) }) }
    fn elem_sub<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.binary(1, a, b) }) }
    fn elem_mul<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.binary(2, a, b) }) }
    fn elem_div<'a>(&'a self, a: &'a GpuTensorHandle, b: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.binary(3, a, b) }) }
    fn unary_sin<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.unary(0, a) }) }
    fn unary_cos<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.unary(1, a) }) }
    fn unary_exp<'a>(&'a self, a: &'a GpuTensorHandle) -> AccelProviderFuture<'a, GpuTensorHandle> { Box::pin(async move { self.unary(12, a) }) }
    // ... the remaining elem_* / unary_* / scalar_* hooks follow the same two-line pattern.

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
    // zeros/ones/fill, reduce_mean(_dim), reduce_min/max, scalar_*, telemetry_snapshot, device_info_struct:
    // same pattern over rmhip_fill / rmhip_reduce / rmhip_scalar / rmhip_telemetry / rmhip_device_info.
}

/// Register before `initialize_acceleration_provider_with` so A/lib.rs:179-181 short-circuits.
pub fn register_hip_provider(device_ordinal: i32) -> Result<()> {
    let provider: &'static HipProvider = Box::leak(Box::new(HipProvider::new(device_ordinal)?));
    unsafe { runmat_accelerate_api::register_provider(provider) }; // lib.rs:3213
    Ok(())
}
