//! UNTESTED - never compiled (the build image has no Rust toolchain).
//!
//! Drop-in content for the zkvm-pairings crate's `src/pairings.rs`, which is an EMPTY file upstream (declared at
//! src/lib.rs:12): `pairing`, `multi_miller_loop`, `final_exponentiation`, `MillerLoopResult`, `Gt` over the crate's own
//! types, computed on MI355X through libzkp_pairings.so (C ABI: include/zkp_pairings.h; raw bindings: the
//! `zkp-pairings-sys` crate in integration/rust/).  Semantics: SURVEY.md S6 / DESIGN.md section 1 - optimal-ate Miller
//! loop over |x|, conjugation for the negative x, final exponentiation f^(3 (p^12 - 1) / r) on the TRUE Frobenius
//! (the crate's own Fp6::frobenius_map, src/fp6.rs:142-176, uses wrong constants).
//!
//! The crate has to grow two accessors, because the point fields are private (src/g1.rs:7-11, src/g2.rs:8-12) and there
//! is no point serialisation.  Add to src/g1.rs and src/g2.rs (inside those modules: only they see the fields):
//!
//! ```ignore
//! impl<C: Curve> G1Affine<C> {
//!     /// (x | y) as 12 canonical limbs + the infinity flag: the wire format of include/zkp_pairings.h
//!     pub fn to_raw(&self) -> ([u64; 12], u8) {
//!         let mut o = [0u64; 12];
//!         o[..6].copy_from_slice(&self.x.0);
//!         o[6..].copy_from_slice(&self.y.0);
//!         (o, self.is_infinity as u8)
//!     }
//!     pub fn from_raw(limbs: &[u64; 12], is_infinity: u8) -> Self {
//!         let f = |i: usize| Fp::<C>::from_raw_unchecked([limbs[i], limbs[i + 1], limbs[i + 2], limbs[i + 3], limbs[i + 4], limbs[i + 5]]);
//!         G1Affine { x: f(0), y: f(6), is_infinity: is_infinity != 0 }
//!     }
//! }
//! impl<C: Curve> G2Affine<C> {
//!     /// (x.c0 | x.c1 | y.c0 | y.c1) as 24 canonical limbs + the infinity flag
//!     pub fn to_raw(&self) -> ([u64; 24], u8) {
//!         let mut o = [0u64; 24];
//!         o[..6].copy_from_slice(&self.x.c0.0);
//!         o[6..12].copy_from_slice(&self.x.c1.0);
//!         o[12..18].copy_from_slice(&self.y.c0.0);
//!         o[18..].copy_from_slice(&self.y.c1.0);
//!         (o, self.is_infinity as u8)
//!     }
//!     pub fn from_raw(limbs: &[u64; 24], is_infinity: u8) -> Self {
//!         let f = |i: usize| Fp::<C>::from_raw_unchecked([limbs[i], limbs[i + 1], limbs[i + 2], limbs[i + 3], limbs[i + 4], limbs[i + 5]]);
//!         G2Affine { x: Fp2 { c0: f(0), c1: f(6) }, y: Fp2 { c0: f(12), c1: f(18) }, is_infinity: is_infinity != 0 }
//!     }
//! }
//! ```
//!
//! and to Cargo.toml: `zkp-pairings-sys = { path = "<this repository>/integration/rust" }`.
//!
//! Two API layers:
//!   * the crate-shaped, context-free functions the north star names - `pairing(&G1Affine, &G2Affine) -> Gt`,
//!     `multi_miller_loop(&[(&G1Affine, &G2Affine)]) -> MillerLoopResult`, `final_exponentiation(&MillerLoopResult) -> Gt`,
//!     `Gt::identity()` - exactly the signatures an upstream `pairings.rs` would export next to src/lib.rs:12.  They run on a
//!     process-global engine that is created on first use (device `ZKP_DEVICE`, default 0) and serialised by a mutex.  A
//!     function of this shape has no error channel: a GPU failure there PANICS with the library's message, the same way the
//!     crate's own arithmetic panics on its `unwrap`s;
//!   * the batched functions on an explicit `Gpu` - what a verifier should call, one launch for thousands of pairings - which
//!     return `Result<_, sys::Error>` and never panic on a GPU error.
use core::marker::PhantomData;

use zkp_pairings_sys as sys;

use crate::common::{Bls12381Curve, Curve};
use crate::fp::Fp;
use crate::fp12::Fp12;
use crate::fp2::Fp2;
use crate::fp6::Fp6;
use crate::g1::G1Affine;
use crate::g2::G2Affine;

// layout the wire format relies on when whole slices are handed over: Fp is six u64 and nothing else (its PhantomData is
// zero-sized), so the tower types are 12 / 36 / 72 u64.  (Conversions below go field by field and do not depend on the
// field ORDER rustc picks for the un-repr(C) structs.)
const _: () = assert!(core::mem::size_of::<Fp<Bls12381Curve>>() == 48);
const _: () = assert!(core::mem::size_of::<Fp2<Bls12381Curve>>() == 96);
const _: () = assert!(core::mem::size_of::<Fp6<Bls12381Curve>>() == 288);
const _: () = assert!(core::mem::size_of::<Fp12<Bls12381Curve>>() == 576);

/// Fp12 -> 72 limbs in declaration order c0.c0.c0, c0.c0.c1, c0.c1.c0 .. c1.c2.c1 (src/fp12.rs:13-16, src/fp6.rs:13-17)
pub fn fp12_to_raw<C: Curve>(f: &Fp12<C>) -> [u64; 72] {
    let mut o = [0u64; 72];
    let c = [&f.c0.c0.c0, &f.c0.c0.c1, &f.c0.c1.c0, &f.c0.c1.c1, &f.c0.c2.c0, &f.c0.c2.c1,
             &f.c1.c0.c0, &f.c1.c0.c1, &f.c1.c1.c0, &f.c1.c1.c1, &f.c1.c2.c0, &f.c1.c2.c1];
    for (i, x) in c.iter().enumerate() {
        o[6 * i..6 * i + 6].copy_from_slice(&x.0);
    }
    o
}

pub fn fp12_from_raw<C: Curve>(l: &[u64]) -> Fp12<C> {
    assert!(l.len() == 72);
    let f = |i: usize| Fp::<C>::from_raw_unchecked([l[6 * i], l[6 * i + 1], l[6 * i + 2], l[6 * i + 3], l[6 * i + 4], l[6 * i + 5]]);
    let f2 = |i: usize| Fp2 { c0: f(i), c1: f(i + 1) };
    let f6 = |i: usize| Fp6 { c0: f2(i), c1: f2(i + 2), c2: f2(i + 4) };
    Fp12 { c0: f6(0), c1: f6(6) }
}

/// The target group element e(P, Q); `Gt::identity()` is `Fp12::one()` (src/fp12.rs:87-89).
#[derive(Clone, Copy)]
pub struct Gt<C: Curve>(pub Fp12<C>);

impl<C: Curve> Gt<C> {
    pub fn identity() -> Self {
        Gt(Fp12::one())
    }
    pub fn is_identity(&self) -> bool {
        *self == Self::identity()
    }
}

impl<C: Curve> PartialEq for Gt<C> {
    fn eq(&self, o: &Self) -> bool {
        fp12_to_raw(&self.0) == fp12_to_raw(&o.0)       // canonical limbs: limb equality is field equality (src/fp12.rs:46-50)
    }
}

/// Output of `multi_miller_loop`, input of `final_exponentiation`.
#[derive(Clone, Copy)]
pub struct MillerLoopResult<C: Curve>(pub Fp12<C>);

impl<C: Curve> MillerLoopResult<C> {
    /// f^(3 (p^12 - 1) / r) on the process-global engine (panics on a GPU error; `Gpu::final_exponentiation_batch` does not)
    pub fn final_exponentiation(&self) -> Gt<C> {
        final_exponentiation(self)
    }
}

/// One GPU (one `zkp_ctx`); create once, reuse for every call - the context owns the device workspace.
pub struct Gpu<C: Curve> {
    eng: sys::Engine,
    _c: PhantomData<C>,
}

fn pack<C: Curve>(terms: &[(&G1Affine<C>, &G2Affine<C>)]) -> (Vec<u64>, Vec<u64>, Vec<u8>, Vec<u8>) {
    let n = terms.len();
    let (mut g1, mut g2) = (Vec::with_capacity(12 * n), Vec::with_capacity(24 * n));
    let (mut i1, mut i2) = (Vec::with_capacity(n), Vec::with_capacity(n));
    for (p, q) in terms {
        let (a, ia) = p.to_raw();
        let (b, ib) = q.to_raw();
        g1.extend_from_slice(&a);
        g2.extend_from_slice(&b);
        i1.push(ia);
        i2.push(ib);
    }
    (g1, g2, i1, i2)
}

fn status_to_result(s: u8) -> Result<(), String> {
    match s {
        0 => Ok(()),
        1 => Err("Point is not on curve".to_string()),          // src/g1.rs:55
        _ => Err("Point is not torsion free".to_string()),      // src/g1.rs:58
    }
}

// ---------------------------------------------------------------------------------------------- batched API: Result, no panics
impl<C: Curve> Gpu<C> {
    /// device = HIP device ordinal of this process
    pub fn new(device: i32) -> Result<Self, sys::Error> {
        Ok(Gpu { eng: sys::Engine::new(device)?, _c: PhantomData })
    }

    /// pairing() of many pairs in one call
    pub fn pairing_batch(&mut self, pairs: &[(&G1Affine<C>, &G2Affine<C>)]) -> Result<Vec<Gt<C>>, sys::Error> {
        let (g1, g2, i1, i2) = pack(pairs);
        let out = self.eng.pairing_batch(&g1, &g2, Some(&i1), Some(&i2))?;
        Ok(out.chunks_exact(72).map(|c| Gt(fp12_from_raw(c))).collect())
    }

    /// `terms.len() / k` Miller loops of k terms each (shared squarings within a group)
    pub fn multi_miller_loop_batch(&mut self, terms: &[(&G1Affine<C>, &G2Affine<C>)], k: usize) -> Result<Vec<MillerLoopResult<C>>, sys::Error> {
        if terms.is_empty() {
            return Ok(Vec::new());
        }
        let (g1, g2, i1, i2) = pack(terms);
        let out = self.eng.multi_miller_loop_batch(&g1, &g2, Some(&i1), Some(&i2), k)?;
        Ok(out.chunks_exact(72).map(|c| MillerLoopResult(fp12_from_raw(c))).collect())
    }

    pub fn final_exponentiation_batch(&mut self, fs: &[MillerLoopResult<C>]) -> Result<Vec<Gt<C>>, sys::Error> {
        let mut raw = Vec::with_capacity(72 * fs.len());
        for f in fs {
            raw.extend_from_slice(&fp12_to_raw(&f.0));
        }
        let out = self.eng.final_exponentiation_batch(&raw)?;
        Ok(out.chunks_exact(72).map(|c| Gt(fp12_from_raw(c))).collect())
    }

    /// The batched form a verifier wants (this is where a GPU pays off): `terms.len() / k` checks of k terms each, check c =
    /// terms [c k, (c + 1) k); returns (flag per check: product of its pairings == Gt::identity(), AND of all flags).
    pub fn pairing_check_batch(&mut self, terms: &[(&G1Affine<C>, &G2Affine<C>)], k: usize) -> Result<(Vec<bool>, bool), sys::Error> {
        let (g1, g2, i1, i2) = pack(terms);
        let (ok, all) = self.eng.pairing_check_batch(&g1, &g2, Some(&i1), Some(&i2), k)?;
        Ok((ok.into_iter().map(|b| b != 0).collect(), all))
    }

    /// `G1Affine::is_valid` / `G2Affine::is_valid` (src/g1.rs:49-62, src/g2.rs:57-69) for many points: per point the crate's
    /// `Result<(), String>` with the crate's own messages; the outer `Result` is the GPU call's
    pub fn g1_is_valid_batch(&mut self, pts: &[&G1Affine<C>]) -> Result<Vec<Result<(), String>>, sys::Error> {
        let mut limbs = Vec::with_capacity(12 * pts.len());
        let mut inf = Vec::with_capacity(pts.len());
        for p in pts {
            let (a, i) = p.to_raw();
            limbs.extend_from_slice(&a);
            inf.push(i);
        }
        Ok(self.eng.g1_is_valid_batch(&limbs, Some(&inf))?.into_iter().map(status_to_result).collect())
    }

    pub fn g2_is_valid_batch(&mut self, pts: &[&G2Affine<C>]) -> Result<Vec<Result<(), String>>, sys::Error> {
        let mut limbs = Vec::with_capacity(24 * pts.len());
        let mut inf = Vec::with_capacity(pts.len());
        for p in pts {
            let (a, i) = p.to_raw();
            limbs.extend_from_slice(&a);
            inf.push(i);
        }
        Ok(self.eng.g2_is_valid_batch(&limbs, Some(&inf))?.into_iter().map(status_to_result).collect())
    }
}

// ---------------------------------------------------------------------------------------------- the crate-shaped, context-free API
// One engine per process, created on first use.  The curve parameter is a compile-time tag only (the engine is BLS12-381,
// like the crate's single `Curve` instance, src/common.rs:62-63), so the global holds the untyped sys::Engine.
static GLOBAL: std::sync::OnceLock<std::sync::Mutex<sys::Engine>> = std::sync::OnceLock::new();

fn with_global<T>(f: impl FnOnce(&mut sys::Engine) -> Result<T, sys::Error>) -> T {
    let m = GLOBAL.get_or_init(|| {
        let dev = std::env::var("ZKP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        std::sync::Mutex::new(sys::Engine::new(dev).unwrap_or_else(|e| panic!("zkvm-pairings: no usable MI355X engine: {e:?}")))
    });
    let mut g = m.lock().unwrap_or_else(|p| p.into_inner());
    f(&mut g).unwrap_or_else(|e| panic!("zkvm-pairings GPU call failed: {e:?}"))
}

/// e(P, Q).  A pair with a point at infinity gives `Gt::identity()`.
pub fn pairing<C: Curve>(p: &G1Affine<C>, q: &G2Affine<C>) -> Gt<C> {
    let (g1, g2, i1, i2) = pack(&[(p, q)]);
    Gt(fp12_from_raw(&with_global(|e| e.pairing_batch(&g1, &g2, Some(&i1), Some(&i2)))))
}

/// prod_i f_{|x|, Q_i}(P_i), conjugated: ONE Miller loop with shared squarings over all terms.
pub fn multi_miller_loop<C: Curve>(terms: &[(&G1Affine<C>, &G2Affine<C>)]) -> MillerLoopResult<C> {
    if terms.is_empty() {
        return MillerLoopResult(Fp12::one());
    }
    let (g1, g2, i1, i2) = pack(terms);
    MillerLoopResult(fp12_from_raw(&with_global(|e| e.multi_miller_loop_batch(&g1, &g2, Some(&i1), Some(&i2), terms.len()))))
}

/// f^(3 (p^12 - 1) / r)
pub fn final_exponentiation<C: Curve>(f: &MillerLoopResult<C>) -> Gt<C> {
    Gt(fp12_from_raw(&with_global(|e| e.final_exponentiation_batch(&fp12_to_raw(&f.0)))))
}

/// The pairing-check shape of the zkVM path: prod_i e(P_i, Q_i) == Gt::identity(), one Miller loop + one final exponentiation
pub fn pairing_check<C: Curve>(terms: &[(&G1Affine<C>, &G2Affine<C>)]) -> bool {
    if terms.is_empty() {
        return true;
    }
    let (g1, g2, i1, i2) = pack(terms);
    with_global(|e| e.pairing_check_batch(&g1, &g2, Some(&i1), Some(&i2), terms.len())).1
}
