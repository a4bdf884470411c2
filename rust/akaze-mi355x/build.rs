fn main() {
    // AKZ_LIB_DIR = directory holding libakz.so (cv_amd/lib of this repository)
    if let Ok(dir) = std::env::var("AKZ_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=akz");
}
