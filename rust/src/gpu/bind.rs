//! Column <-> Target binding and the batch fill.  Source only (see rust/README.md).
//!
//! While the circuit is built, every hot-path generator's output targets are recorded in registration order -- the
//! order `p2e_schedule_describe(0, ..)` lists, checked in the shipping repository against two independent walks of
//! the gadgets (tests/test_host.py, tests/test_check_circuit.py):
//!   NonNativeAddition / Subtraction (gadgets/nonnative.rs:254,365)   sum.value.limbs[0..9], overflow.target
//!   NonNativeMultipleAdds (:323)                                     sum.value.limbs[0..9], overflow.0
//!   NonNativeInverse (:511)                                          inv.limbs[0..9], div.limbs[0..9]
//!   MulNonnativeGate row, CheckSumGate row (:396-449)                wires r(0..9), q(0..9), check_sum(0..17), then b(0..16)
//!   GLVDecomposition (gadgets/glv.rs:67)                             k1.limbs[0..5], k2.limbs[0..5], k1_neg, k2_neg
use std::ffi::CStr;

use anyhow::{ensure, Result};
use num::BigUint;
use plonky2::field::types::{Field, PrimeField};
use plonky2::hash::hash_types::RichField;
use plonky2::iop::target::Target;
use plonky2::iop::witness::{PartialWitness, WitnessWrite};

use super::ffi::*;

/// Output targets of the 3 555 hot-path generators, in registration order (len == 82 615).
pub struct HotPathBinding {
    pub targets: Vec<Target>,
}

/// One signature's inputs as the circuit sees them (gadgets/ecdsa.rs:30-36).
pub struct VerifyInput {
    pub msg: BigUint,
    pub r: BigUint,
    pub s: BigUint,
    pub pk_x: BigUint,
    pub pk_y: BigUint,
}

fn pack32(vals: impl Iterator<Item = BigUint>, n: usize) -> Vec<u8> {
    let mut out = vec![0u8; 32 * n];
    for (i, v) in vals.enumerate() {
        let b = v.to_bytes_le();
        out[32 * i..32 * i + b.len()].copy_from_slice(&b); // values are < 2^256 (from_noncanonical_biguint would panic otherwise)
    }
    out
}

/// Pre-seeds one `PartialWitness` per signature with every hot-path generator output, so that
/// `generate_partial_witness` finds them set (a generator's `run_once` may then start with
/// `if witness.contains_all(&outputs) { return Ok(()) }`).  `ctx` was created with `P2E_CTX_HOST_POINTERS`.
pub fn fill_partial_witnesses<F: RichField>(
    binding: &HotPathBinding,
    ctx: *mut P2eCtx,
    sigs: &[VerifyInput],
    pws: &mut [PartialWitness<F>],
) -> Result<()> {
    let n = sigs.len();
    ensure!(binding.targets.len() == P2E_VERIFY_COLS && pws.len() == n);
    let msg = pack32(sigs.iter().map(|s| s.msg.clone()), n);
    let r = pack32(sigs.iter().map(|s| s.r.clone()), n);
    let s = pack32(sigs.iter().map(|s| s.s.clone()), n);
    let px = pack32(sigs.iter().map(|s| s.pk_x.clone()), n);
    let py = pack32(sigs.iter().map(|s| s.pk_y.clone()), n);
    let mut cols = vec![0u64; P2E_VERIFY_COLS * n]; // column-major over the batch: cols[c * n + i]
    let (mut err, mut valid) = (vec![0u8; n], vec![0u8; n]);
    let rc = unsafe {
        p2e_ecdsa_verify_witness_batch(ctx, msg.as_ptr(), r.as_ptr(), s.as_ptr(), px.as_ptr(), py.as_ptr(), cols.as_mut_ptr(),
                                       n, n, err.as_mut_ptr(), valid.as_mut_ptr())
    };
    ensure!(rc >= 0, "p2e: {}", unsafe { CStr::from_ptr(p2e_last_error()) }.to_string_lossy());
    for i in 0..n {
        // where err != 0 the reference generators panic (inverse of zero, limb range ...): surface it as Err
        ensure!(err[i] == 0, "signature {i}: witness generation error bits {:#x}", err[i]);
        for (c, t) in binding.targets.iter().enumerate() {
            pws[i].set_target(*t, F::from_canonical_u64(cols[c * n + i]))?;
        }
    }
    Ok(())
}

/// The wire map of `p2e_assemble_wires` from the same binding: a target that is a gate wire lands at
/// `wire * degree + row` of the proof's wire matrix; virtual targets are reached through the copies plonky2
/// records for them and are left to `generate_partial_witness`.
pub fn wire_map_of(binding: &HotPathBinding, degree: usize) -> Vec<P2eWireMapEntry> {
    binding
        .targets
        .iter()
        .enumerate()
        .filter_map(|(c, t)| match t {
            Target::Wire(w) => Some(P2eWireMapEntry { src: P2E_WIRE_SRC_COLS | c as u32, dst: (w.column * degree + w.row) as u32 }),
            Target::VirtualTarget { .. } => None,
        })
        .collect()
}

/// One built circuit of `verify_p256_message_circuit` (gadgets/ecdsa.rs:55-78): its hot-path output targets in
/// registration order (len == 115 557 = `p2e_curve_program_num_cols`) and the library's program object, which carries
/// the point `precompute_window` drew with `rand()` while THIS circuit was built (gadgets/curve_windowed_mul.rs:57).
pub struct P256Binding {
    pub targets: Vec<Target>,
    pub program: *mut P2eCurveProgram,
}

/// `blind` = that point (the builder wrapper kept it).  Call once per built circuit; free with
/// `p2e_curve_program_destroy`.
pub fn bind_p256_verifier(ctx: *mut P2eCtx, targets: Vec<Target>, blind_x: &BigUint, blind_y: &BigUint) -> Result<P256Binding> {
    let (bx, by) = (pack32(std::iter::once(blind_x.clone()), 1), pack32(std::iter::once(blind_y.clone()), 1));
    let mut program = std::ptr::null_mut();
    let rc = unsafe { p2e_curve_program_create(ctx, P2E_CP_VERIFY, P2E_CURVE_P256, bx.as_ptr(), by.as_ptr(), &mut program) };
    ensure!(rc == 0, "p2e: {}", unsafe { CStr::from_ptr(p2e_last_error()) }.to_string_lossy());
    ensure!(targets.len() as i64 == unsafe { p2e_curve_program_num_cols(program) });
    Ok(P256Binding { targets, program })
}

/// The P-256 counterpart of `fill_partial_witnesses` (same pre-seeding, same error mapping).
pub fn fill_partial_witnesses_p256<F: RichField>(
    binding: &P256Binding,
    ctx: *mut P2eCtx,
    sigs: &[VerifyInput],
    pws: &mut [PartialWitness<F>],
) -> Result<()> {
    let n = sigs.len();
    let ncols = binding.targets.len();
    ensure!(pws.len() == n);
    let msg = pack32(sigs.iter().map(|s| s.msg.clone()), n);
    let r = pack32(sigs.iter().map(|s| s.r.clone()), n);
    let s = pack32(sigs.iter().map(|s| s.s.clone()), n);
    let px = pack32(sigs.iter().map(|s| s.pk_x.clone()), n);
    let py = pack32(sigs.iter().map(|s| s.pk_y.clone()), n);
    let mut cols = vec![0u64; ncols * n];
    let (mut err, mut valid) = (vec![0u8; n], vec![0u8; n]);
    let rc = unsafe {
        p2e_p256_verify_witness_batch(ctx, binding.program, msg.as_ptr(), r.as_ptr(), s.as_ptr(), px.as_ptr(), py.as_ptr(),
                                      cols.as_mut_ptr(), n, n, err.as_mut_ptr(), valid.as_mut_ptr())
    };
    ensure!(rc >= 0, "p2e: {}", unsafe { CStr::from_ptr(p2e_last_error()) }.to_string_lossy());
    for i in 0..n {
        ensure!(err[i] == 0, "signature {i}: witness generation error bits {:#x}", err[i]);
        for (c, t) in binding.targets.iter().enumerate() {
            pws[i].set_target(*t, F::from_canonical_u64(cols[c * n + i]))?;
        }
    }
    Ok(())
}
