// build.rs addition for zlogic/matrix-eyes (Cargo feature `hip`): links libmatrixeyes_hip.so.
// NOT COMPILED in the build image (no Rust toolchain there); see INTEGRATION.md.
//
// Cargo.toml:
//   [features]
//   hip = []      # MI355X back end; no burn backend feature is needed beside it
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_ok() {
        let dir = std::env::var("MATRIX_EYES_HIP_DIR")
            .expect("MATRIX_EYES_HIP_DIR = directory that holds libmatrixeyes_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=matrixeyes_hip");
        println!("cargo:rerun-if-env-changed=MATRIX_EYES_HIP_DIR");
    }
}
