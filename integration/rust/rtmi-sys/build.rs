// librtmi.so leaves the HIP runtime undefined on purpose (one HIP runtime per process), so the
// final binary links both. RTMI_LIB_DIR = directory holding librtmi.so, ROCM_PATH = ROCm prefix.
fn main() {
    let lib_dir = std::env::var("RTMI_LIB_DIR").unwrap_or_else(|_| "../../../raytracing_amd".into());
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}", lib_dir);
    println!("cargo:rustc-link-lib=dylib=rtmi");
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-env-changed=RTMI_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
