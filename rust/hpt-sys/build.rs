// Links libhpt.so (built by `make -C path-tracer_amd`, hipcc --offload-arch=gfx950).  HPT_LIB_DIR overrides where
// the library lives; HPT_BUILD=1 runs the Makefile first.
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let dir = env::var("HPT_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../../path-tracer_amd"));
    if env::var("HPT_BUILD").map(|v| v == "1").unwrap_or(false) {
        let status = Command::new("make").arg("-C").arg(&dir).arg("libhpt.so").status().expect("could not run make");
        assert!(status.success(), "building libhpt.so failed");
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=hpt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-changed={}", manifest.join("../../include/hpt.h").display());
    println!("cargo:rerun-if-env-changed=HPT_LIB_DIR");
}
