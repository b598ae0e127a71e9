// Tells rustc where libjubjub_hip.so lives: JUBJUB_HIP_LIB_DIR, or the in-tree build directory of this repository
// (python -m jubjub_amd.build writes jubjub_amd/lib/libjubjub_hip.so).  The library itself links the HIP runtime; nothing else is needed here.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("JUBJUB_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("jubjub_amd").join("lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=JUBJUB_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=src/ffi.rs");
}
