// Builds libmi355pt.so from the repository this crate lives in (hipcc cross-compiles gfx950 code
// objects without a GPU) and links it.  MI355PT_DIR overrides the repository root.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = env::var("MI355PT_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../..")
    });
    let csrc = root.join("pathtracer_amd/csrc");
    let status = Command::new("make")
        .arg("-C")
        .arg(&csrc)
        .arg("../libmi355pt.so")
        .status()
        .expect("failed to run make");
    assert!(status.success(), "building libmi355pt.so failed");
    let libdir = root.join("pathtracer_amd");
    println!("cargo:rustc-link-search=native={}", libdir.display());
    println!("cargo:rustc-link-lib=dylib=mi355pt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", libdir.display());
    println!("cargo:rerun-if-changed={}", root.join("include/mi355pt.h").display());
    println!("cargo:rerun-if-changed={}", csrc.display());
}
