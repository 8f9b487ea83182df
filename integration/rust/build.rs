// UNTESTED (no Rust toolchain in the build image).  Links the in-tree shared library.
fn main() {
    let dir = std::env::var("ZKP_PAIRINGS_LIB_DIR").unwrap_or_else(|_| "../../zkvm_pairings_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=zkp_pairings");
}
