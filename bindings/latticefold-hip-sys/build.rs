// Links liblfhip.so (built by `make -C latticefold_amd/csrc`, gfx950 only).  LFHIP_LIB_DIR overrides the in-tree location.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("LFHIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../latticefold_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=lfhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=LFHIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/lfhip.h");
    println!("cargo:rerun-if-changed=../../include/lfplus.h");
}
