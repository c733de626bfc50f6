// UNBUILT SOURCE (no Rust toolchain in this image).  Builds librtx950.so for gfx950 with hipcc from the
// sources of this repository and links it; RTX950_SRC points at raytracing_amd/csrc (default: the path
// relative to this crate inside the repository), HIPCC at the compiler (default /opt/rocm/bin/hipcc).
// The flags are those of raytracing_amd/csrc/Makefile: -ffp-contract=off is part of the exactness contract.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let src = env::var("RTX950_SRC")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../raytracing_amd/csrc"));
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let lib = out.join("librtx950.so");
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared", "-o"])
        .arg(&lib)
        .arg(src.join("rtx950_kernels.hip"))
        .args(["-x", "hip"])
        .arg(src.join("rtx950_api.cpp"))
        .status()
        .expect("failed to run hipcc");
    assert!(status.success(), "hipcc failed");
    for f in ["rtx950_kernels.hip", "rtx950_api.cpp", "rtx950_device.h"] {
        println!("cargo:rerun-if-changed={}", src.join(f).display());
    }
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rtx950");
}
