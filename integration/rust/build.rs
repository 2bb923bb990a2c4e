// Link against libcorrla_rsvd.so.  CORRLA_RSVD_LIB_DIR names the directory that holds it (in this repository:
// corrla_rs_amd/lib, built by `python -m corrla_rs_amd.build`).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("CORRLA_RSVD_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../corrla_rs_amd/lib"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=corrla_rsvd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=CORRLA_RSVD_LIB_DIR");
}
