//! wiring.rs -- the reference-side edits that make `provider = "hip"` selectable, as code rather than prose.
//!
//! NOT compiled here (no Rust toolchain in the build image).  Three small edits to the reference, each shown as the items a
//! maintainer adds; everything else of RunMat is untouched.  `shim/hip_provider.rs` + `shim/rmhip_sys.rs` go to
//! `crates/runmat-accelerate/src/backend/hip/{mod.rs, rmhip_sys.rs}`.
//!
//! 1. crates/runmat-accelerate/Cargo.toml
//!        [features]
//!        hip = []                       # links librmhip.so; no Rust dependencies
//!    crates/runmat-accelerate/build.rs (new, or appended)
//!        if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
//!            if let Some(dir) = std::env::var_os("RMHIP_LIB_DIR") {
//!                println!("cargo:rustc-link-search=native={}", std::path::Path::new(&dir).display());
//!            }
//!            println!("cargo:rustc-link-lib=dylib=rmhip");
//!        }
//!
//! 2. The preference enums gain one arm each - `crates/runmat-accelerate/src/lib.rs:47-53` and its configuration twin
//!    `crates/runmat-config/src/runtime/schema/accelerate.rs:31-39` (`[accelerate] provider = "hip"`, `--accel-provider hip`):

/// crates/runmat-accelerate/src/lib.rs:47-53 with the new arm
#[derive(Debug, Clone, Copy, PartialEq, Eq, Serialize, Deserialize)]
#[serde(rename_all = "kebab-case")]
pub enum AccelerateProviderPreference {
    Auto,
    Wgpu,
    InProcess,
    /// AMD Instinct (gfx950) through librmhip.so; `Auto` tries it first when the crate is built with `--features hip`
    Hip,
}

/// crates/runmat-config/src/runtime/schema/accelerate.rs:31-39 with the new arm (serde / clap names: "hip")
#[derive(Debug, Clone, Copy, PartialEq, Eq, Serialize, Deserialize, ValueEnum)]
#[serde(rename_all = "kebab-case")]
pub enum ConfigAccelerateProviderPreference {
    Auto,
    Wgpu,
    #[serde(rename = "inprocess", alias = "in-process")]
    #[value(name = "inprocess")]
    InProcess,
    Hip,
}

// 3. `initialize_acceleration_provider_with` (crates/runmat-accelerate/src/lib.rs:176-259): one block in front of the wgpu block.
//    The function already returns early when a provider is registered (:179-181) and falls through to the in-process provider when
//    nothing registered (:247-258), so a failing rmhip_init (no gfx950 device: RMHIP_ERR_NO_DEVICE) degrades exactly like a failing
//    wgpu adapter does.

#[cfg(feature = "hip")]
mod backend_hip_registration {
    use super::{AccelerateInitOptions, AccelerateProviderPreference};
    use crate::backend::hip::{register_hip_provider, HipProvider};

    /// Called at the top of the `let registered = { ... }` block; `true` ends the search.
    pub(super) fn try_register(options: &AccelerateInitOptions) -> bool {
        if !matches!(options.provider, AccelerateProviderPreference::Auto | AccelerateProviderPreference::Hip) {
            return false;
        }
        // one provider per process = one GPU: multi-GPU jobs run one process per device and pick it here
        let ordinal = std::env::var("RUNMAT_HIP_DEVICE").ok().and_then(|v| v.parse::<i32>().ok()).unwrap_or(0);
        match register_hip_provider(ordinal) {
            Ok(()) => {
                // `runmat accel-info` / telemetry read these through the trait: device_info_struct(), telemetry_snapshot()
                if let Some(p) = runmat_accelerate_api::provider() {
                    let info = p.device_info_struct();
                    log::info!(
                        "RunMat Accelerate: using HIP provider '{}' (vendor {}, backend {}, {} bytes)",
                        info.name,
                        info.vendor,
                        info.backend.as_deref().unwrap_or("?"),
                        info.memory_bytes.unwrap_or(0)
                    );
                }
                true
            }
            Err(err) => {
                if matches!(options.provider, AccelerateProviderPreference::Hip) {
                    log::warn!("RunMat Accelerate: failed to initialize the HIP provider, falling back: {err}");
                }
                false
            }
        }
    }
    #[allow(dead_code)]
    fn _type_check(_: &HipProvider) {}
}

// in initialize_acceleration_provider_with, before the wgpu block:
//
//     #[cfg(feature = "hip")]
//     if backend_hip_registration::try_register(options) {
//         return;
//     }
//
// Auto-offload: the thresholds the planner compares against come from `AutoOffloadOptions` / `RUNMAT_ACCEL_THRESHOLD_*`
// (native_auto.rs:841-1060); `profiles/r03_offload_calibration.json` is a calibration sample of this backend in the schema
// `apply_auto_offload_calibration_from_file` loads (native_auto.rs:330-420), produced by tests/tools/offload_calibrate.cpp.
