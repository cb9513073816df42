// UNVERIFIED (no Rust toolchain in the build image).  Links libzg_halo2.so when the `zg` feature is on.
// ZG_HALO2_LIB_DIR = the directory that holds libzg_halo2.so (the repo's 0g-halo2_amd/ after `make`).
fn main() {
    println!("cargo:rerun-if-env-changed=ZG_HALO2_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_ZG").is_some() {
        let dir = std::env::var("ZG_HALO2_LIB_DIR")
            .expect("set ZG_HALO2_LIB_DIR to the directory that holds libzg_halo2.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=zg_halo2");
    }
}
