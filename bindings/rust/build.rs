fn main() {
    let dir = std::env::var("MINAVERIFY_LIB_DIR").unwrap_or_else(|_| "../../mina_bridge_amd".into());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=minaverify");
}
