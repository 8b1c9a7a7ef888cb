// Links the prebuilt shared library (make -C atoma-infer_amd => atoma-infer_amd/lib/libatoma_hip.so) and the HIP runtime --
// the counterpart of the reference's `cargo:rustc-link-lib=static=flashattention` + `dylib=cudart` (csrc/build.rs:105-113).
// ATOMA_HIP_LIB_DIR overrides the search path; ROCM_PATH defaults to /opt/rocm.
use std::{env, path::PathBuf};

fn main() {
    let lib_dir = env::var("ATOMA_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../atoma-infer_amd/lib")
    });
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=atoma_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-env-changed=ATOMA_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/atoma_hip.h");
}
