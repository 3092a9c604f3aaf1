// Links librtamd.so.  RTAMD_LIB_DIR names the directory that holds it (default: the
// rust-raytracer_amd directory beside this crate, where `make -C rust-raytracer_amd` puts it).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("RTAMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("rust-raytracer_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rtamd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RTAMD_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/rtamd.h");
}
