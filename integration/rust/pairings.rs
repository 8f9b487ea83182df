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
    pub fn final_exponentiation(&self, gpu: &mut Gpu<C>) -> Gt<C> {
        final_exponentiation(gpu, self)
    }
}

/// One GPU (one `zkp_ctx`); create once, reuse for every call - the context owns the device workspace.
pub struct Gpu<C: Curve> {
    eng: sys::Engine,
    _c: PhantomData<C>,
}

impl<C: Curve> Gpu<C> {
    /// device = HIP device ordinal of this process
    pub fn new(device: i32) -> Result<Self, sys::Error> {
        Ok(Gpu { eng: sys::Engine::new(device)?, _c: PhantomData })
    }
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

/// e(P, Q).  A pair with a point at infinity gives `Gt::identity()`.
pub fn pairing<C: Curve>(gpu: &mut Gpu<C>, p: &G1Affine<C>, q: &G2Affine<C>) -> Gt<C> {
    let (g1, g2, i1, i2) = pack(&[(p, q)]);
    let out = gpu.eng.pairing_batch(&g1, &g2, Some(&i1), Some(&i2)).expect("zkp_pairing_batch");
    Gt(fp12_from_raw(&out))
}

/// prod_i f_{|x|, Q_i}(P_i), conjugated: ONE Miller loop with shared squarings over all terms.
pub fn multi_miller_loop<C: Curve>(gpu: &mut Gpu<C>, terms: &[(&G1Affine<C>, &G2Affine<C>)]) -> MillerLoopResult<C> {
    if terms.is_empty() {
        return MillerLoopResult(Fp12::one());
    }
    let (g1, g2, i1, i2) = pack(terms);
    let out = gpu.eng.multi_miller_loop_batch(&g1, &g2, Some(&i1), Some(&i2), terms.len()).expect("zkp_multi_miller_loop_batch");
    MillerLoopResult(fp12_from_raw(&out))
}

/// f^(3 (p^12 - 1) / r)
pub fn final_exponentiation<C: Curve>(gpu: &mut Gpu<C>, f: &MillerLoopResult<C>) -> Gt<C> {
    let out = gpu.eng.final_exponentiation_batch(&fp12_to_raw(&f.0)).expect("zkp_final_exponentiation_batch");
    Gt(fp12_from_raw(&out))
}

/// The batched form a verifier wants (this is where a GPU pays off): `checks.len() / k` checks of k terms each,
/// check c = terms [c k, (c + 1) k); returns (flag per check: product of its pairings == Gt::identity(), AND of all flags).
pub fn pairing_check_batch<C: Curve>(gpu: &mut Gpu<C>, terms: &[(&G1Affine<C>, &G2Affine<C>)], k: usize) -> (Vec<bool>, bool) {
    assert!(k > 0 && terms.len() % k == 0);
    let (g1, g2, i1, i2) = pack(terms);
    let (ok, all) = gpu.eng.pairing_check_batch(&g1, &g2, Some(&i1), Some(&i2), k).expect("zkp_pairing_check_batch");
    (ok.into_iter().map(|b| b != 0).collect(), all)
}

/// pairing() of many pairs in one call
pub fn pairing_batch<C: Curve>(gpu: &mut Gpu<C>, pairs: &[(&G1Affine<C>, &G2Affine<C>)]) -> Vec<Gt<C>> {
    let (g1, g2, i1, i2) = pack(pairs);
    let out = gpu.eng.pairing_batch(&g1, &g2, Some(&i1), Some(&i2)).expect("zkp_pairing_batch");
    out.chunks_exact(72).map(|c| Gt(fp12_from_raw(c))).collect()
}

/// `G1Affine::is_valid` / `G2Affine::is_valid` (src/g1.rs:49-62, src/g2.rs:57-69) for many points: the crate's
/// `Result<(), String>` with the crate's own messages
pub fn g1_is_valid_batch<C: Curve>(gpu: &mut Gpu<C>, pts: &[&G1Affine<C>]) -> Vec<Result<(), String>> {
    let mut limbs = Vec::with_capacity(12 * pts.len());
    let mut inf = Vec::with_capacity(pts.len());
    for p in pts {
        let (a, i) = p.to_raw();
        limbs.extend_from_slice(&a);
        inf.push(i);
    }
    gpu.eng.g1_is_valid_batch(&limbs, Some(&inf)).expect("zkp_g1_is_valid_batch").into_iter().map(status_to_result).collect()
}

pub fn g2_is_valid_batch<C: Curve>(gpu: &mut Gpu<C>, pts: &[&G2Affine<C>]) -> Vec<Result<(), String>> {
    let mut limbs = Vec::with_capacity(24 * pts.len());
    let mut inf = Vec::with_capacity(pts.len());
    for p in pts {
        let (a, i) = p.to_raw();
        limbs.extend_from_slice(&a);
        inf.push(i);
    }
    gpu.eng.g2_is_valid_batch(&limbs, Some(&inf)).expect("zkp_g2_is_valid_batch").into_iter().map(status_to_result).collect()
}

fn status_to_result(s: u8) -> Result<(), String> {
    match s {
        0 => Ok(()),
        1 => Err("Point is not on curve".to_string()),          // src/g1.rs:55
        _ => Err("Point is not torsion free".to_string()),      // src/g1.rs:58
    }
}
