//! UNTESTED (no Rust toolchain in the build image).  Raw bindings to include/zkp_pairings.h plus the safe
//! batch wrappers a zkvm-pairings maintainer would call.  Wire formats are the crate's own in-memory
//! layouts: `Fp.0: [u64; 6]` canonical little-endian limbs (reference src/fp.rs:24), Fp12 in declaration
//! order (src/fp12.rs:13-16), points as coordinate arrays plus a parallel infinity byte array.
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct ZkpCtx {
    _private: [u8; 0],
}

extern "C" {
    pub fn zkp_abi_version() -> c_int;
    pub fn zkp_strerror(status: c_int) -> *const c_char;
    pub fn zkp_init(device: c_int, out_ctx: *mut *mut ZkpCtx) -> c_int;
    pub fn zkp_free(ctx: *mut ZkpCtx);
    pub fn zkp_last_error(ctx: *const ZkpCtx) -> *const c_char;
    pub fn zkp_set_validate(ctx: *mut ZkpCtx, on: c_int) -> c_int;
    pub fn zkp_gt_identity() -> *const u64;
    pub fn zkp_pairing_batch(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8, inf2: *const u8,
                             n: usize, out_gt: *mut u64) -> c_int;
    pub fn zkp_multi_miller_loop_batch(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8,
                                       inf2: *const u8, n_checks: usize, k: usize, out_ml: *mut u64) -> c_int;
    pub fn zkp_final_exponentiation_batch(ctx: *mut ZkpCtx, f: *const u64, n: usize, out_gt: *mut u64) -> c_int;
    pub fn zkp_pairing_check_batch(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8,
                                   inf2: *const u8, n_checks: usize, k: usize, ok: *mut u8, all_ok: *mut c_int) -> c_int;
    /// the whole batch as ONE product check: Miller product over all n pairs, Fp12 product, one final exponentiation
    pub fn zkp_miller_product(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8, inf2: *const u8,
                              n: usize, out_ml: *mut u64) -> c_int;
    pub fn zkp_fp12_product(ctx: *mut ZkpCtx, f: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn zkp_pairing_product_check(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8,
                                     inf2: *const u8, n: usize, out_gt: *mut u64, is_one: *mut c_int) -> c_int;
    pub fn zkp_g1_is_valid_batch(ctx: *mut ZkpCtx, g1: *const u64, inf: *const u8, n: usize, status: *mut u8) -> c_int;
    pub fn zkp_g2_is_valid_batch(ctx: *mut ZkpCtx, g2: *const u64, inf: *const u8, n: usize, status: *mut u8) -> c_int;
    pub fn zkp_pairing_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                 d_inf2: *const c_void, n: usize, d_out_gt: *mut c_void, stream: *mut c_void) -> c_int;
}

/// One GPU = one engine.  Not `Sync`: a `zkp_ctx` is not thread-safe.
pub struct Engine(*mut ZkpCtx);

impl Engine {
    pub fn new(device: i32) -> Result<Self, i32> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { zkp_init(device, &mut p) };
        if rc == 0 { Ok(Engine(p)) } else { Err(rc) }
    }

    /// `g1`: n x 12 limbs (x | y), `g2`: n x 24 limbs (x.c0 | x.c1 | y.c0 | y.c1); returns n x 72 limbs of Gt.
    pub fn pairing_batch(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>) -> Result<Vec<u64>, i32> {
        let n = g1.len() / 12;
        assert_eq!(g2.len(), 24 * n);
        let mut out = vec![0u64; 72 * n];
        let rc = unsafe {
            zkp_pairing_batch(self.0, g1.as_ptr(), g2.as_ptr(), inf1.map_or(core::ptr::null(), |s| s.as_ptr()),
                              inf2.map_or(core::ptr::null(), |s| s.as_ptr()), n, out.as_mut_ptr())
        };
        if rc == 0 { Ok(out) } else { Err(rc) }
    }

    /// n_checks products of k pairings each against `Gt::identity()`; returns (per-check flags, AND of all).
    pub fn pairing_check_batch(&mut self, g1: &[u64], g2: &[u64], k: usize) -> Result<(Vec<u8>, bool), i32> {
        let n = g1.len() / 12;
        assert!(k > 0 && n % k == 0 && g2.len() == 24 * n);
        let mut ok = vec![0u8; n / k];
        let mut all = 1;
        let rc = unsafe {
            zkp_pairing_check_batch(self.0, g1.as_ptr(), g2.as_ptr(), core::ptr::null(), core::ptr::null(), n / k, k,
                                    ok.as_mut_ptr(), &mut all)
        };
        if rc == 0 { Ok((ok, all != 0)) } else { Err(rc) }
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { zkp_free(self.0) }
    }
}
