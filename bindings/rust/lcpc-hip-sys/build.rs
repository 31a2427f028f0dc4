// Links liblcpc_hip.so (built by `make -C lcpc_amd/csrc`, hipcc --offload-arch=gfx950) and the HIP runtime.
//   LCPC_HIP_LIB_DIR   directory holding liblcpc_hip.so   (default: <repo>/lcpc_amd/lib, three levels above this crate)
//   ROCM_PATH          ROCm installation                    (default: /opt/rocm)
// librccl is NOT linked: the library dlopen()s it when a sharded encoder asks for a communicator.
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let lib_dir = env::var("LCPC_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("../../../lcpc_amd/lib"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=lcpc_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-env-changed=LCPC_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed=../../../include/lcpc_hip.h");
}
