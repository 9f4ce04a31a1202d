// Builds the HIP library with hipcc for gfx950 (through its Makefile) and links it.
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let root = PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
    let csrc = root.join("beat_detector_amd").join("csrc");
    let status = Command::new("make")
        .arg("-C")
        .arg(&csrc)
        .status()
        .expect("make (hipcc from ROCm must be installed)");
    assert!(status.success(), "building libbeatdet_hip.so failed");
    println!("cargo:rustc-link-search=native={}", root.join("beat_detector_amd").display());
    println!("cargo:rustc-link-lib=dylib=beatdet_hip");
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-changed={}", root.join("include").join("beat_detector_hip.h").display());
}
