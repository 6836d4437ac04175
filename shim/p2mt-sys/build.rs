// Links libp2mt_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"` at the repository root).
// P2MT_LIB_DIR overrides the default location (<repo>/plonky2-merkle-trees_amd).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("P2MT_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("..").join("plonky2-merkle-trees_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=p2mt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=P2MT_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/p2mt.h");
}
