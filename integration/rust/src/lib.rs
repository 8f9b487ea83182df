//! UNTESTED - the build image has no Rust toolchain; this crate has never been compiled.
//!
//! `zkp-pairings-sys`: raw bindings to include/zkp_pairings.h (ABI version 4) plus safe batch wrappers over slices of
//! limbs.  Wire formats are the zkvm-pairings crate's own in-memory layouts: `Fp.0: [u64; 6]` canonical little-endian
//! limbs (reference src/fp.rs:24), Fp12 in declaration order (src/fp12.rs:13-16), points as coordinate arrays plus a
//! parallel infinity byte array.  The crate-level API (`pairing`, `multi_miller_loop`, `final_exponentiation`, `Gt`)
//! over the crate's own types is `integration/rust/pairings.rs`, the file to drop into the crate's empty src/pairings.rs.
use core::ffi::{c_char, c_int, c_uint, c_void};
use core::ptr;

#[repr(C)]
pub struct ZkpCtx {
    _private: [u8; 0],
}

pub const ZKP_OK: c_int = 0;
pub const ZKP_ERR_ARG: c_int = -1;
pub const ZKP_ERR_NO_DEVICE: c_int = -2;
pub const ZKP_ERR_HIP: c_int = -3;
pub const ZKP_ERR_NONCANONICAL: c_int = -4;
pub const ZKP_ERR_OOM: c_int = -5;
pub const ZKP_ERR_COMM: c_int = -6;
/// bytes of the communicator id `zkp_comm_unique_id` produces (RCCL's ncclUniqueId)
pub const ZKP_COMM_ID_BYTES: usize = 128;
/// zkp_point_status of `zkp_points_check_batch`
pub const ZKP_POINT_OK: u8 = 0;
pub const ZKP_POINT_NONCANONICAL: u8 = 1;
pub const ZKP_POINT_MALFORMED: u8 = 2;
pub const ZKP_POINT_NOT_ON_CURVE: u8 = 3;
pub const ZKP_POINT_NOT_IN_SUBGROUP: u8 = 4;

/// zkp_fp_op: 0 and 1 are the zkVM precompile's op numbers (reference src/fp.rs:376,443)
pub const ZKP_FP_MUL: c_int = 0;
pub const ZKP_FP_ADD: c_int = 1;
pub const ZKP_FP_SUB: c_int = 2;
pub const ZKP_FP_NEG: c_int = 3;
pub const ZKP_FP_SQUARE: c_int = 4;
pub const ZKP_FP_INVERT: c_int = 5;
pub const ZKP_FP_CORE28: c_int = 16;

extern "C" {
    pub fn zkp_abi_version() -> c_int;
    pub fn zkp_strerror(status: c_int) -> *const c_char;
    pub fn zkp_init(device: c_int, out_ctx: *mut *mut ZkpCtx) -> c_int;
    pub fn zkp_free(ctx: *mut ZkpCtx);
    pub fn zkp_last_error(ctx: *const ZkpCtx) -> *const c_char;
    pub fn zkp_set_validate(ctx: *mut ZkpCtx, on: c_int) -> c_int;
    pub fn zkp_set_kernel(ctx: *mut ZkpCtx, kind: c_int) -> c_int;
    pub fn zkp_device_info(ctx: *const ZkpCtx, cus: *mut c_int, clock_khz: *mut c_int, name: *mut c_char, name_len: usize) -> c_int;
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
    pub fn zkp_g1_mul_batch(ctx: *mut ZkpCtx, base: *const u64, base_stride: usize, scalars: *const u64, n: usize,
                            out: *mut u64, out_inf: *mut u8) -> c_int;
    pub fn zkp_g2_mul_batch(ctx: *mut ZkpCtx, base: *const u64, base_stride: usize, scalars: *const u64, n: usize,
                            out: *mut u64, out_inf: *mut u8) -> c_int;
    pub fn zkp_g1_decode_batch(ctx: *mut ZkpCtx, bytes: *const u8, n: usize, out_g1: *mut u64, out_inf: *mut u8, status: *mut u8) -> c_int;
    pub fn zkp_g2_decode_batch(ctx: *mut ZkpCtx, bytes: *const u8, n: usize, out_g2: *mut u64, out_inf: *mut u8, status: *mut u8) -> c_int;
    pub fn zkp_g1_encode_batch(ctx: *mut ZkpCtx, g1: *const u64, inf: *const u8, n: usize, out_bytes: *mut u8) -> c_int;
    pub fn zkp_g2_encode_batch(ctx: *mut ZkpCtx, g2: *const u64, inf: *const u8, n: usize, out_bytes: *mut u8) -> c_int;
    pub fn zkp_fp_op_batch(ctx: *mut ZkpCtx, op: c_int, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn zkp_tower_op_batch(ctx: *mut ZkpCtx, op: c_int, a: *const u64, b: *const u64, n: usize, repeat: u32, out: *mut u64) -> c_int;

    /// several GPUs behind one call: contiguous blocks of checks per context, AND of the flags on the host
    pub fn zkp_pairing_check_batch_multi(ctxs: *const *mut ZkpCtx, n_ctx: c_int, g1: *const u64, g2: *const u64, inf1: *const u8,
                                         inf2: *const u8, n_checks: usize, k: usize, ok: *mut u8, all_ok: *mut c_int) -> c_int;
    pub fn zkp_pairing_batch_multi(ctxs: *const *mut ZkpCtx, n_ctx: c_int, g1: *const u64, g2: *const u64, inf1: *const u8,
                                   inf2: *const u8, n: usize, out_gt: *mut u64, ok: *mut u8, all_ok: *mut c_int) -> c_int;

    // page-locked host memory: from it the host-pointer entry points' copies are DMAs that overlap the kernels
    pub fn zkp_host_alloc(bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn zkp_host_free(ptr: *mut c_void) -> c_int;
    pub fn zkp_host_register(ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn zkp_host_unregister(ptr: *mut c_void) -> c_int;

    // device-pointer flavours (buffers resident in HBM, asynchronous on a hipStream_t passed as *mut c_void)
    pub fn zkp_pairing_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                 d_inf2: *const c_void, n: usize, d_out_gt: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_multi_miller_loop_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                           d_inf2: *const c_void, n_checks: usize, k: usize, d_out_ml: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_final_exponentiation_batch_dev(ctx: *mut ZkpCtx, d_f: *const c_void, n: usize, d_out_gt: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_pairing_check_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                       d_inf2: *const c_void, n_checks: usize, k: usize, d_ok: *mut c_void, d_all_ok: *mut c_void,
                                       stream: *mut c_void) -> c_int;
    pub fn zkp_pairing_gt_check_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                          d_inf2: *const c_void, n_checks: usize, k: usize, d_out_gt: *mut c_void, d_ok: *mut c_void,
                                          d_all_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_miller_product_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                  d_inf2: *const c_void, n: usize, d_out_ml: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_fp12_product_dev(ctx: *mut ZkpCtx, d_f: *const c_void, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_pairing_product_check_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                         d_inf2: *const c_void, n: usize, d_out_gt: *mut c_void, d_is_one: *mut c_void,
                                         stream: *mut c_void) -> c_int;
    pub fn zkp_g1_is_valid_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_inf: *const c_void, n: usize, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_g2_is_valid_batch_dev(ctx: *mut ZkpCtx, d_g2: *const c_void, d_inf: *const c_void, n: usize, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_g1_mul_batch_dev(ctx: *mut ZkpCtx, d_base: *const c_void, base_stride: usize, d_scalars: *const c_void, n: usize,
                                d_out: *mut c_void, d_out_inf: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_g2_mul_batch_dev(ctx: *mut ZkpCtx, d_base: *const c_void, base_stride: usize, d_scalars: *const c_void, n: usize,
                                d_out: *mut c_void, d_out_inf: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_g1_decode_batch_dev(ctx: *mut ZkpCtx, d_bytes: *const c_void, n: usize, d_out_g1: *mut c_void, d_out_inf: *mut c_void,
                                   d_status: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_g2_decode_batch_dev(ctx: *mut ZkpCtx, d_bytes: *const c_void, n: usize, d_out_g2: *mut c_void, d_out_inf: *mut c_void,
                                   d_status: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_g1_encode_batch_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_inf: *const c_void, n: usize, d_out_bytes: *mut c_void,
                                   stream: *mut c_void) -> c_int;
    pub fn zkp_g2_encode_batch_dev(ctx: *mut ZkpCtx, d_g2: *const c_void, d_inf: *const c_void, n: usize, d_out_bytes: *mut c_void,
                                   stream: *mut c_void) -> c_int;
    /// BASELINE config 5 in one call: raw uncompressed points -> Fp::from_bytes, is_valid, pairing check (status bytes + flags)
    pub fn zkp_points_check_batch(ctx: *mut ZkpCtx, g1_bytes: *const u8, g2_bytes: *const u8, n_checks: usize, k: usize, st1: *mut u8,
                                  st2: *mut u8, ok: *mut u8, all_ok: *mut c_int) -> c_int;
    pub fn zkp_points_check_batch_dev(ctx: *mut ZkpCtx, d_g1_bytes: *const c_void, d_g2_bytes: *const c_void, n_checks: usize, k: usize,
                                      d_st1: *mut c_void, d_st2: *mut c_void, d_ok: *mut c_void, d_all_ok: *mut c_void,
                                      stream: *mut c_void) -> c_int;
    /// one rank per GPU: the path's ONE collective (RCCL all-reduce(MIN) of the AND flag) behind the ABI
    pub fn zkp_comm_unique_id(out_id: *mut c_void) -> c_int;
    pub fn zkp_comm_init_rank(ctx: *mut ZkpCtx, nranks: c_int, rank: c_int, unique_id: *const c_void) -> c_int;
    pub fn zkp_comm_destroy(ctx: *mut ZkpCtx) -> c_int;
    pub fn zkp_comm_info(ctx: *const ZkpCtx, nranks: *mut c_int, rank: *mut c_int) -> c_int;
    pub fn zkp_and_allreduce_dev(ctx: *mut ZkpCtx, d_flag: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_pairing_check_batch_allreduce(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8, inf2: *const u8,
                                             n_checks: usize, k: usize, ok: *mut u8, all_ok: *mut c_int) -> c_int;
    pub fn zkp_pairing_check_batch_allreduce_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                                 d_inf2: *const c_void, n_checks: usize, k: usize, d_ok: *mut c_void,
                                                 d_all_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_pairing_gt_check_batch_allreduce_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, d_inf1: *const c_void,
                                                    d_inf2: *const c_void, n_checks: usize, k: usize, d_out_gt: *mut c_void,
                                                    d_ok: *mut c_void, d_all_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_points_check_batch_allreduce(ctx: *mut ZkpCtx, g1_bytes: *const u8, g2_bytes: *const u8, n_checks: usize, k: usize, st1: *mut u8,
                                            st2: *mut u8, ok: *mut u8, all_ok: *mut c_int) -> c_int;
    pub fn zkp_points_check_batch_allreduce_dev(ctx: *mut ZkpCtx, d_g1_bytes: *const c_void, d_g2_bytes: *const c_void, n_checks: usize,
                                                k: usize, d_st1: *mut c_void, d_st2: *mut c_void, d_ok: *mut c_void,
                                                d_all_ok: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn zkp_pairing_product_check_allgather(ctx: *mut ZkpCtx, g1: *const u64, g2: *const u64, inf1: *const u8, inf2: *const u8,
                                               n: usize, out_gt: *mut u64, is_one: *mut c_int) -> c_int;
    pub fn zkp_take_validation_status_dev(ctx: *mut ZkpCtx, stream: *mut c_void, bad: *mut c_int) -> c_int;
    pub fn zkp_clock_probe_dev(ctx: *mut ZkpCtx, stream: *mut c_void, spin_us: c_uint, d_out: *mut c_void, wall_khz: *mut c_int) -> c_int;
    pub fn zkp_profile_pairing_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, n: usize, d_out_gt: *mut c_void,
                                   ms: *mut f32, launches: *mut c_int) -> c_int;
    pub fn zkp_time_pairing_dev(ctx: *mut ZkpCtx, d_g1: *const c_void, d_g2: *const c_void, n: usize, d_out_gt: *mut c_void,
                                reps: c_int, avg_ms: *mut f32) -> c_int;
    pub fn zkp_time_coop_step(ctx: *mut ZkpCtx, which: c_int, n: usize, ms: *mut f32) -> c_int;
}

/// status of a call that failed: the negative `zkp_status` and what the library has to say about it
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Error {
    pub status: i32,
    pub detail: String,
}

/// Page-locked host memory (`zkp_host_alloc`): the container to keep point / Gt arrays in when they are handed to the
/// host-pointer entry points - their copies are then DMAs that overlap the kernels (and, in the multi-GPU call, the other
/// GPUs' copies) instead of blocking staged copies.  Derefs to a slice; usable with every engine of the process.
pub struct PinnedVec<T: Pod> {
    ptr: *mut T,
    len: usize,
}
/// The element types a `PinnedVec` may hold: plain integers, for which the all-zero bit pattern is a value (a `Copy` bound alone
/// would admit references and `NonZero*`, whose zeroed form is undefined behaviour reachable from safe code).  Sealed.
pub trait Pod: Copy + sealed::Sealed {}
mod sealed {
    pub trait Sealed {}
    impl Sealed for u8 {}
    impl Sealed for u32 {}
    impl Sealed for u64 {}
    impl Sealed for i32 {}
}
impl Pod for u8 {}
impl Pod for u32 {}
impl Pod for u64 {}
impl Pod for i32 {}
unsafe impl<T: Pod + Send> Send for PinnedVec<T> {}

impl<T: Pod> PinnedVec<T> {
    pub fn zeroed(len: usize) -> Result<Self, Error> {
        let mut p: *mut c_void = core::ptr::null_mut();
        let bytes = match len.checked_mul(core::mem::size_of::<T>()) {
            Some(b) => core::cmp::max(1, b),
            None => return Err(Error { status: ZKP_ERR_ARG, detail: format!("PinnedVec of {len} elements overflows usize") }),
        };
        let rc = unsafe { zkp_host_alloc(bytes, &mut p) };
        if rc != ZKP_OK {
            return Err(Error { status: rc, detail: format!("zkp_host_alloc({bytes} bytes)") });
        }
        unsafe { core::ptr::write_bytes(p as *mut u8, 0, bytes) };
        Ok(PinnedVec { ptr: p as *mut T, len })
    }
    pub fn from_slice(s: &[T]) -> Result<Self, Error> {
        let mut v = Self::zeroed(s.len())?;
        v.copy_from_slice(s);
        Ok(v)
    }
}
impl<T: Pod> core::ops::Deref for PinnedVec<T> {
    type Target = [T];
    fn deref(&self) -> &[T] {
        unsafe { core::slice::from_raw_parts(self.ptr, self.len) }
    }
}
impl<T: Pod> core::ops::DerefMut for PinnedVec<T> {
    fn deref_mut(&mut self) -> &mut [T] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}
impl<T: Pod> Drop for PinnedVec<T> {
    fn drop(&mut self) {
        unsafe { zkp_host_free(self.ptr as *mut c_void) };
    }
}

/// One GPU = one engine.  `Send` but not `Sync`: a `zkp_ctx` is not thread-safe.
pub struct Engine(*mut ZkpCtx);
unsafe impl Send for Engine {}

fn opt_ptr(s: Option<&[u8]>) -> *const u8 {
    s.map_or(core::ptr::null(), |s| s.as_ptr())
}

impl Engine {
    pub fn new(device: i32) -> Result<Self, Error> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { zkp_init(device, &mut p) };
        if rc == ZKP_OK { Ok(Engine(p)) } else { Err(Error { status: rc, detail: format!("zkp_init(device = {device})") }) }
    }

    pub fn as_ptr(&self) -> *mut ZkpCtx {
        self.0
    }

    fn err(&self, rc: c_int) -> Error {
        let c = unsafe { core::ffi::CStr::from_ptr(zkp_last_error(self.0)) };
        Error { status: rc, detail: c.to_string_lossy().into_owned() }
    }

    /// lengths of a pair batch: `g1` n x 12 limbs, `g2` n x 24 limbs, the optional infinity arrays n bytes each.
    /// Every wrapper checks them BEFORE the FFI call: the C side trusts its size arguments.
    fn pairs(g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>) -> Result<usize, Error> {
        let bad = |what: &str| Err(Error { status: ZKP_ERR_ARG, detail: what.to_string() });
        if g1.len() % 12 != 0 {
            return bad("g1 is not a whole number of points (12 limbs each)");
        }
        let n = g1.len() / 12;
        if g2.len() != 24 * n {
            return bad("g2 does not hold one point (24 limbs) per g1 point");
        }
        if inf1.map_or(false, |s| s.len() != n) || inf2.map_or(false, |s| s.len() != n) {
            return bad("an infinity array does not hold one byte per pair");
        }
        Ok(n)
    }

    /// `g1`: n x 12 limbs (x | y), `g2`: n x 24 limbs (x.c0 | x.c1 | y.c0 | y.c1); returns n x 72 limbs of Gt.
    pub fn pairing_batch(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>) -> Result<Vec<u64>, Error> {
        let n = Self::pairs(g1, g2, inf1, inf2)?;
        let mut out = vec![0u64; 72 * n];
        let rc = unsafe { zkp_pairing_batch(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n, out.as_mut_ptr()) };
        if rc == ZKP_OK { Ok(out) } else { Err(self.err(rc)) }
    }

    /// the same into a caller-owned output (e.g. a `PinnedVec<u64>` of 72 n limbs): no allocation, DMA both ways
    pub fn pairing_batch_into(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>, out_gt: &mut [u64]) -> Result<(), Error> {
        let n = Self::pairs(g1, g2, inf1, inf2)?;
        if out_gt.len() != 72 * n {
            return Err(Error { status: ZKP_ERR_ARG, detail: "out_gt does not hold 72 limbs per pair".into() });
        }
        let rc = unsafe { zkp_pairing_batch(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n, out_gt.as_mut_ptr()) };
        if rc == ZKP_OK { Ok(()) } else { Err(self.err(rc)) }
    }

    /// n / k Miller loop values (72 limbs each) of groups of k consecutive pairs
    pub fn multi_miller_loop_batch(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>, k: usize) -> Result<Vec<u64>, Error> {
        let n = Self::pairs(g1, g2, inf1, inf2)?;
        if k == 0 || n % k != 0 {
            return Err(Error { status: ZKP_ERR_ARG, detail: "the number of pairs is not a multiple of k".into() });
        }
        let mut out = vec![0u64; 72 * (n / k)];
        let rc = unsafe { zkp_multi_miller_loop_batch(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n / k, k, out.as_mut_ptr()) };
        if rc == ZKP_OK { Ok(out) } else { Err(self.err(rc)) }
    }

    pub fn final_exponentiation_batch(&mut self, f: &[u64]) -> Result<Vec<u64>, Error> {
        if f.len() % 72 != 0 {
            return Err(Error { status: ZKP_ERR_ARG, detail: "f is not a whole number of Fp12 values (72 limbs each)".into() });
        }
        let mut out = vec![0u64; f.len()];
        let rc = unsafe { zkp_final_exponentiation_batch(self.0, f.as_ptr(), f.len() / 72, out.as_mut_ptr()) };
        if rc == ZKP_OK { Ok(out) } else { Err(self.err(rc)) }
    }

    /// n / k products of k pairings each against `Gt::identity()`; returns (per-check flags, AND of all).
    pub fn pairing_check_batch(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>, k: usize) -> Result<(Vec<u8>, bool), Error> {
        let n = Self::pairs(g1, g2, inf1, inf2)?;
        if k == 0 || n % k != 0 {
            return Err(Error { status: ZKP_ERR_ARG, detail: "the number of pairs is not a multiple of k".into() });
        }
        let mut ok = vec![0u8; n / k];
        let mut all: c_int = 1;
        let rc = unsafe { zkp_pairing_check_batch(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n / k, k, ok.as_mut_ptr(), &mut all) };
        if rc == ZKP_OK { Ok((ok, all != 0)) } else { Err(self.err(rc)) }
    }

    /// prod_i e(P_i, Q_i) == Gt::identity() with ONE final exponentiation; returns (Gt, is_one)
    pub fn pairing_product_check(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>) -> Result<([u64; 72], bool), Error> {
        let n = Self::pairs(g1, g2, inf1, inf2)?;
        let mut gt = [0u64; 72];
        let mut one: c_int = 0;
        let rc = unsafe { zkp_pairing_product_check(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n, gt.as_mut_ptr(), &mut one) };
        if rc == ZKP_OK { Ok((gt, one != 0)) } else { Err(self.err(rc)) }
    }

    /// status per point: 0 valid (or infinity), 1 not on curve, 2 not torsion free (reference src/g1.rs:49-62, src/g2.rs:57-69)
    pub fn g1_is_valid_batch(&mut self, g1: &[u64], inf: Option<&[u8]>) -> Result<Vec<u8>, Error> {
        if g1.len() % 12 != 0 || inf.map_or(false, |s| s.len() != g1.len() / 12) {
            return Err(Error { status: ZKP_ERR_ARG, detail: "g1 / inf lengths".into() });
        }
        let mut st = vec![0u8; g1.len() / 12];
        let rc = unsafe { zkp_g1_is_valid_batch(self.0, g1.as_ptr(), opt_ptr(inf), st.len(), st.as_mut_ptr()) };
        if rc == ZKP_OK { Ok(st) } else { Err(self.err(rc)) }
    }

    pub fn g2_is_valid_batch(&mut self, g2: &[u64], inf: Option<&[u8]>) -> Result<Vec<u8>, Error> {
        if g2.len() % 24 != 0 || inf.map_or(false, |s| s.len() != g2.len() / 24) {
            return Err(Error { status: ZKP_ERR_ARG, detail: "g2 / inf lengths".into() });
        }
        let mut st = vec![0u8; g2.len() / 24];
        let rc = unsafe { zkp_g2_is_valid_batch(self.0, g2.as_ptr(), opt_ptr(inf), st.len(), st.as_mut_ptr()) };
        if rc == ZKP_OK { Ok(st) } else { Err(self.err(rc)) }
    }

    /// BASELINE config 5 in one call: `g1_bytes` n x 96, `g2_bytes` n x 192 uncompressed big-endian points, groups of k pairs.
    /// Returns (status per G1 point, status per G2 point [`ZKP_POINT_*`], per-check flags, AND of all): a check passes when all
    /// its points decode, are valid (reference `is_valid`, src/g1.rs:49-62, src/g2.rs:57-69) and its pairing product is the identity.
    #[allow(clippy::type_complexity)]
    pub fn points_check_batch(&mut self, g1_bytes: &[u8], g2_bytes: &[u8], k: usize) -> Result<(Vec<u8>, Vec<u8>, Vec<u8>, bool), Error> {
        if g1_bytes.len() % 96 != 0 || g2_bytes.len() != 2 * g1_bytes.len() || k == 0 || (g1_bytes.len() / 96) % k != 0 {
            return Err(Error { status: ZKP_ERR_ARG, detail: "byte string lengths / k".into() });
        }
        let n = g1_bytes.len() / 96;
        let (mut st1, mut st2, mut ok) = (vec![0u8; n], vec![0u8; n], vec![0u8; n / k]);
        let mut all: c_int = 1;
        let rc = unsafe {
            zkp_points_check_batch(self.0, g1_bytes.as_ptr(), g2_bytes.as_ptr(), n / k, k, st1.as_mut_ptr(), st2.as_mut_ptr(), ok.as_mut_ptr(), &mut all)
        };
        if rc == ZKP_OK { Ok((st1, st2, ok, all != 0)) } else { Err(self.err(rc)) }
    }

    /// Join the job's communicator (one rank per GPU; `id` from `comm_unique_id()` on rank 0, handed to every rank by the host's
    /// own means).  Collective: returns when all `nranks` ranks have called it.
    pub fn comm_init_rank(&mut self, nranks: i32, rank: i32, id: &[u8; ZKP_COMM_ID_BYTES]) -> Result<(), Error> {
        let rc = unsafe { zkp_comm_init_rank(self.0, nranks, rank, id.as_ptr() as *const c_void) };
        if rc == ZKP_OK { Ok(()) } else { Err(self.err(rc)) }
    }

    /// This rank's contiguous block of a sharded check + the path's ONE collective (RCCL all-reduce(MIN) of the AND flag):
    /// returns (this rank's per-check flags, AND over ALL ranks' checks).  Every rank must call it once per global check.
    ///
    /// A LOCAL argument error (slice lengths, k) must not keep this rank out of the collective its peers are waiting in: the
    /// wrappers below then enter it through the same entry point with arguments the library itself refuses (no points, one
    /// check) - the rank takes part with flag 0 / the zero record, every rank reads `false` - and return the local error.
    pub fn pairing_check_batch_allreduce(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>, k: usize)
                                         -> Result<(Vec<u8>, bool), Error> {
        let n = match Self::pairs(g1, g2, inf1, inf2) {
            Ok(n) if k != 0 && n % k == 0 => n,
            other => {
                let mut all: c_int = 0;
                unsafe { zkp_pairing_check_batch_allreduce(self.0, ptr::null(), ptr::null(), ptr::null(), ptr::null(), 1, 1, ptr::null_mut(), &mut all) };
                return Err(other.err().unwrap_or(Error { status: ZKP_ERR_ARG, detail: "the number of pairs is not a multiple of k".into() }));
            }
        };
        let mut ok = vec![0u8; n / k];
        let mut all: c_int = 1;
        let rc = unsafe {
            zkp_pairing_check_batch_allreduce(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n / k, k, ok.as_mut_ptr(), &mut all)
        };
        if rc == ZKP_OK { Ok((ok, all != 0)) } else { Err(self.err(rc)) }
    }

    /// config 5 on a node: this rank's block of `points_check_batch` + the AND over all ranks
    #[allow(clippy::type_complexity)]
    pub fn points_check_batch_allreduce(&mut self, g1_bytes: &[u8], g2_bytes: &[u8], k: usize) -> Result<(Vec<u8>, Vec<u8>, Vec<u8>, bool), Error> {
        if g1_bytes.len() % 96 != 0 || g2_bytes.len() != 2 * g1_bytes.len() || k == 0 || (g1_bytes.len() / 96) % k != 0 {
            let mut all: c_int = 0;
            unsafe { zkp_points_check_batch_allreduce(self.0, ptr::null(), ptr::null(), 1, 1, ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), &mut all) };
            return Err(Error { status: ZKP_ERR_ARG, detail: "byte string lengths / k".into() });
        }
        let n = g1_bytes.len() / 96;
        let (mut st1, mut st2, mut ok) = (vec![0u8; n], vec![0u8; n], vec![0u8; n / k]);
        let mut all: c_int = 1;
        let rc = unsafe {
            zkp_points_check_batch_allreduce(self.0, g1_bytes.as_ptr(), g2_bytes.as_ptr(), n / k, k, st1.as_mut_ptr(), st2.as_mut_ptr(), ok.as_mut_ptr(), &mut all)
        };
        if rc == ZKP_OK { Ok((st1, st2, ok, all != 0)) } else { Err(self.err(rc)) }
    }

    /// ONE product check over the whole sharded batch: this rank's Miller product, one all-gather of 576 B per rank, one final
    /// exponentiation; every rank gets the same (Gt, is_one)
    pub fn pairing_product_check_allgather(&mut self, g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>) -> Result<([u64; 72], bool), Error> {
        let mut gt = [0u64; 72];
        let mut one: c_int = 0;
        let n = match Self::pairs(g1, g2, inf1, inf2) {
            Ok(n) => n,
            Err(e) => {
                unsafe { zkp_pairing_product_check_allgather(self.0, ptr::null(), ptr::null(), ptr::null(), ptr::null(), 1, ptr::null_mut(), &mut one) };
                return Err(e);
            }
        };
        let rc = unsafe { zkp_pairing_product_check_allgather(self.0, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n, gt.as_mut_ptr(), &mut one) };
        if rc == ZKP_OK { Ok((gt, one != 0)) } else { Err(self.err(rc)) }
    }
}

/// the 128-byte communicator id (rank 0 makes it; every rank passes the same bytes to `Engine::comm_init_rank`)
pub fn comm_unique_id() -> Result<[u8; ZKP_COMM_ID_BYTES], Error> {
    let mut id = [0u8; ZKP_COMM_ID_BYTES];
    let rc = unsafe { zkp_comm_unique_id(id.as_mut_ptr() as *mut c_void) };
    if rc == ZKP_OK { Ok(id) } else { Err(Error { status: rc, detail: "zkp_comm_unique_id".into() }) }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { zkp_free(self.0) }
    }
}

/// Several GPUs from one host thread: one `Engine` per GPU, contiguous blocks of checks, flags ANDed on the host.
pub fn pairing_check_batch_multi(engines: &mut [Engine], g1: &[u64], g2: &[u64], inf1: Option<&[u8]>, inf2: Option<&[u8]>, k: usize)
                                 -> Result<(Vec<u8>, bool), Error> {
    let n = Engine::pairs(g1, g2, inf1, inf2)?;
    if engines.is_empty() || engines.len() > 64 || k == 0 || n % k != 0 {
        return Err(Error { status: ZKP_ERR_ARG, detail: "engines / k".into() });
    }
    let ptrs: Vec<*mut ZkpCtx> = engines.iter().map(|e| e.0).collect();
    let mut ok = vec![0u8; n / k];
    let mut all: c_int = 1;
    let rc = unsafe {
        zkp_pairing_check_batch_multi(ptrs.as_ptr(), ptrs.len() as c_int, g1.as_ptr(), g2.as_ptr(), opt_ptr(inf1), opt_ptr(inf2), n / k, k,
                                      ok.as_mut_ptr(), &mut all)
    };
    if rc == ZKP_OK { Ok((ok, all != 0)) } else { Err(engines[0].err(rc)) }
}
