// NEVER COMPILED HERE.  native/build.rs for the hip backend: the WGSL -> SPIR-V step of the reference
// (native/build.rs:5-36) disappears with native/shaders/*.wgsl; the kernels live in libp3hip.so
// (make -C plonky3-mobile_amd/csrc, hipcc --offload-arch=gfx950) and are only linked here.
use std::env;

fn main() {
    // directory holding libp3hip.so (default: the in-tree build output of this repository)
    let dir = env::var("P3HIP_LIB_DIR").unwrap_or_else(|_| "../plonky3-mobile_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=p3hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=P3HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/p3hip.h");
}
