// Drives hipcc for gfx950 only; there is no other backend and no fallback.
// -ffp-contract=off is part of the numeric contract, not an optimisation choice.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../..");
    let csrc = root.join("renderer_amd/csrc");
    let lib = out.join("libsxe_hip.so");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let sources = ["sxe_api.hip", "sxe_geometry.hip", "sxe_raster.hip"];
    let status = Command::new(hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"])
        .arg("-o")
        .arg(&lib)
        .args(sources.iter().map(|f| csrc.join(f)))
        .status()
        .expect("hipcc not found");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=sxe_hip");
    for f in sources.iter().chain(["sxe_internal.h", "sxe_device.h", "sxe_math.h", "sxe_srgb_tables.h"].iter()) {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", root.join("include/sxe.h").display());
}
