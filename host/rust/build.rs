// build.rs of the reference crate: builds the HIP library with hipcc (through its Makefile) and links it.
// STARKHIP_DIR = checkout of the repository that holds stark_verifier_amd/csrc.
// Source only: never compiled in the repository it ships in (no Rust toolchain there).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("STARKHIP_DIR").expect("STARKHIP_DIR = checkout of the starkhip repository"));
    let status = Command::new("make")
        .arg("-C")
        .arg(root.join("stark_verifier_amd/csrc"))
        .arg("../libstarkhip.so")
        .status()
        .expect("make / hipcc not found");
    assert!(status.success(), "building libstarkhip.so failed");
    let lib_dir = root.join("stark_verifier_amd");
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=starkhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-env-changed=STARKHIP_DIR");
}
