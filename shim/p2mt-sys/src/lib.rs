//! `extern "C"` block of libp2mt_hip.so (include/p2mt.h), generated: see ../../src/ffi.rs.
#[path = "../../src/ffi.rs"]
mod ffi;
pub use ffi::*;

/// Status code -> the reference's error convention (panic), with the library's message.
pub fn ok(rc: i32) {
    if rc != P2MT_OK {
        let msg = unsafe { std::ffi::CStr::from_ptr(p2mt_last_error()) }.to_string_lossy().into_owned();
        panic!("p2mt status {rc}: {msg}");
    }
}
