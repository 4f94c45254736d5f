// Links libpp_hip.so (built by `python -m path_planner_amd.build`).
// PP_HIP_LIB_DIR names the directory that holds it; by default the in-tree build directory.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("PP_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap())
            .join("../../path_planner_amd/csrc")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=pp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PP_HIP_LIB_DIR");
}
