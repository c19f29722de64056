//! `extern "C"` surface of libp2e_hip.so, mirroring include/p2e.h declaration by declaration.
//! Every function cites, in the header, the reference `run_once` body or gadget it stands in for.
use core::ffi::{c_char, c_void};

#[repr(C)]
pub struct P2eCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct P2eWireMap {
    _private: [u8; 0],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct P2eGenDesc {
    pub kind: i32, // 0 add, 1 sub, 2 add_many, 3 mul(+checksum), 4 inv, 5 glv_decomposition
    pub field: i32, // 0 Secp256K1Base, 1 Secp256K1Scalar
    pub first_col: u32,
    pub num_cols: u32,
    pub label: [u8; 48],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct P2eGenWiring {
    pub num_operands: i32,
    pub src: [u32; 4],
    pub num_limbs: [u8; 4],
    pub range_check: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct P2eAuxDesc {
    pub kind: i32,
    pub first_col: u32,
    pub num_cols: u32,
    pub label: [u8; 48],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct P2eUxDesc {
    pub first_col: u32,
    pub num_cols: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct P2eWireMapEntry {
    pub src: u32, // P2E_WIRE_SRC_{COLS,AUX,UX} | column
    pub dst: u32, // wire * degree + row
}

/// a block of witness columns completed by one launch of a fused call (include/p2e.h p2e_segments_describe)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct P2eSegmentDesc {
    pub first_col: u32,
    pub num_cols: u32,
}

pub const P2E_CTX_HOST_POINTERS: u32 = 1;
pub const P2E_CTX_ASYNC: u32 = 2;
pub const P2E_CTX_PHASE_TIMING: u32 = 4;
pub const P2E_VERIFY_COLS: usize = 82_615;
pub const P2E_VERIFY_AUX_COLS: usize = 8_959;
pub const P2E_VERIFY_UX_COLS: usize = 249_385;
pub const P2E_SRC_AUX: u32 = 0x2000_0000;
pub const P2E_SRC_INPUT: u32 = 0x4000_0000;
pub const P2E_SRC_CONST: u32 = 0x8000_0000;
pub const P2E_WIRE_SRC_COLS: u32 = 0x0000_0000;
pub const P2E_WIRE_SRC_AUX: u32 = 0x4000_0000;
pub const P2E_WIRE_SRC_UX: u32 = 0x8000_0000;
pub const P2E_WIRE_SRC_GATE: u32 = 0xC000_0000;
pub const P2E_COMPACT_WIDE: u32 = 0x8000_0000;

/// one built circuit of a curve program (include/p2e.h p2e_curve_program)
#[repr(C)]
pub struct P2eCurveProgram {
    _private: [u8; 0],
}
pub const P2E_CURVE_SECP256K1: i32 = 0;
pub const P2E_CURVE_P256: i32 = 1;
pub const P2E_CP_WINDOWED_MUL: i32 = 1;
pub const P2E_CP_SCALAR_MUL: i32 = 2;
pub const P2E_CP_VERIFY: i32 = 3;

// tests/test_ffi_surface.py parses this block and include/p2e.h and requires the same symbols, argument counts and
// integer widths: add a declaration here whenever the header gains one.
#[link(name = "p2e_hip")]
extern "C" {
    // ---- context
    pub fn p2e_ctx_create(device: i32, flags: u32, stream: *mut c_void, out: *mut *mut P2eCtx) -> i32;
    pub fn p2e_ctx_destroy(ctx: *mut P2eCtx);
    pub fn p2e_sync(ctx: *mut P2eCtx) -> i32;
    pub fn p2e_last_error() -> *const c_char;
    pub fn p2e_scratch_bytes(program: i32, n: usize) -> usize;
    // per-phase / per-kernel HIP-event timings of the last fused call on this context (bench.py's roofline figures)
    pub fn p2e_last_phase_ms(ctx: *mut P2eCtx, out: *mut f32, cap: i32) -> i32;
    // column blocks of the last fused call as they become final: what an RCCL exchange / D2H copy / prover can start on
    pub fn p2e_segments_describe(ctx: *mut P2eCtx, out: *mut P2eSegmentDesc, cap: usize) -> i64;
    pub fn p2e_segment_stream_wait(ctx: *mut P2eCtx, segment: i32, stream: *mut c_void) -> i32;
    pub fn p2e_segment_sync(ctx: *mut P2eCtx, segment: i32) -> i32;

    // ---- the fused schedules: every hot-path run_once of one circuit instance per batch element
    // gadgets/ecdsa.rs:30-53 (gates/mul_nonnative.rs:249-324,513-531; gadgets/nonnative.rs:626-645,696-728,792-810,
    // 857-872; gadgets/glv.rs:128-142)
    pub fn p2e_ecdsa_verify_witness_batch(ctx: *mut P2eCtx, msg32: *const u8, r32: *const u8, s32: *const u8, pkx32: *const u8,
        pky32: *const u8, cols: *mut u64, n: usize, ld: usize, err: *mut u8, valid: *mut u8) -> i64;
    pub fn p2e_ecdsa_verify_witness_compact_batch(ctx: *mut P2eCtx, msg32: *const u8, r32: *const u8, s32: *const u8,
        pkx32: *const u8, pky32: *const u8, narrow: *mut u32, ld_narrow: usize, wide: *mut u64, ld_wide: usize, n: usize,
        err: *mut u8, valid: *mut u8) -> i64;
    pub fn p2e_glv_mul_witness_batch(ctx: *mut P2eCtx, px32: *const u8, py32: *const u8, k32: *const u8, cols: *mut u64,
        n: usize, ld: usize, err: *mut u8, valid: *mut u8) -> i64;
    pub fn p2e_glv_mul_witness_compact_batch(ctx: *mut P2eCtx, px32: *const u8, py32: *const u8, k32: *const u8,
        narrow: *mut u32, ld_narrow: usize, wide: *mut u64, ld_wide: usize, n: usize, err: *mut u8, valid: *mut u8) -> i64;
    // the circuit's verdict alone (curve/ecdsa.rs:42-62 verify_message with the circuit's semantics)
    pub fn p2e_ecdsa_verify_batch(ctx: *mut P2eCtx, msg32: *const u8, r32: *const u8, s32: *const u8, pkx32: *const u8,
        pky32: *const u8, n: usize, err: *mut u8, valid: *mut u8) -> i64;

    // ---- single generators (one run_once body each)
    pub fn p2e_mul_witness_batch(ctx: *mut P2eCtx, field: i32, x: *const u64, y: *const u64, r: *mut u64, q: *mut u64,
        check_sum: *mut u64, b: *mut u64, n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_checksum_witness_batch(ctx: *mut P2eCtx, a: *const u64, b: *mut u64, n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_add_witness_batch(ctx: *mut P2eCtx, field: i32, a: *const u64, b: *const u64, sum: *mut u64, overflow: *mut u64,
        n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_sub_witness_batch(ctx: *mut P2eCtx, field: i32, a: *const u64, b: *const u64, diff: *mut u64, overflow: *mut u64,
        n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_add_many_witness_batch(ctx: *mut P2eCtx, field: i32, summands: *const u64, k: i32, sum: *mut u64,
        overflow: *mut u64, n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_inv_witness_batch(ctx: *mut P2eCtx, field: i32, x: *const u64, inv: *mut u64, div: *mut u64, n: usize, ld: usize,
        err: *mut u8) -> i64;
    pub fn p2e_biguint_div_rem_batch(ctx: *mut P2eCtx, a: *const u64, na: i32, b: *const u64, nb: i32, div: *mut u64,
        rem: *mut u64, n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_glv_decompose_batch(ctx: *mut P2eCtx, k: *const u64, k1: *mut u64, k2: *mut u64, k1_neg: *mut u64,
        k2_neg: *mut u64, n: usize, ld: usize, err: *mut u8) -> i64;
    pub fn p2e_limb_split(ctx: *mut P2eCtx, packed: *const u8, limbs: *mut u64, n: usize, ld: usize) -> i64;
    pub fn p2e_limb_pack(ctx: *mut P2eCtx, limbs: *const u64, packed: *mut u8, n: usize, ld: usize, err: *mut u8) -> i64;

    // ---- the targets other generators fill on the same path
    pub fn p2e_aux_witness_batch(ctx: *mut P2eCtx, program: i32, pky32: *const u8, cols: *const u64, ld: usize, aux: *mut u64,
        ld_aux: usize, n: usize, err: *mut u8) -> i64;
    pub fn p2e_aux_witness_compact_batch(ctx: *mut P2eCtx, program: i32, pky32: *const u8, narrow: *const u32,
        ld_narrow: usize, aux32: *mut u32, ld_aux: usize, n: usize, err: *mut u8) -> i64;
    pub fn p2e_aux_describe(program: i32, out: *mut P2eAuxDesc, cap: usize) -> i64;
    pub fn p2e_aux_num_cols(program: i32) -> i64;
    pub fn p2e_ux_witness_batch(ctx: *mut P2eCtx, program: i32, msg32: *const u8, r32: *const u8, s32: *const u8,
        pkx32: *const u8, pky32: *const u8, cols: *const u64, ld: usize, aux: *const u64, ld_aux: usize, ux: *mut c_void,
        ux_u32: i32, ld_ux: usize, n: usize, err: *mut u8) -> i64;
    pub fn p2e_ux_describe(program: i32, out: *mut P2eUxDesc, cap: usize) -> i64;
    pub fn p2e_ux_num_cols(program: i32) -> i64;

    // ---- wire-matrix assembly (rank 3)
    pub fn p2e_wire_map_create(ctx: *mut P2eCtx, program: i32, entries: *const P2eWireMapEntry, count: usize, num_wires: u32,
        degree: u32, out: *mut *mut P2eWireMap) -> i32;
    pub fn p2e_wire_map_destroy(ctx: *mut P2eCtx, map: *mut P2eWireMap);
    pub fn p2e_assemble_wires(ctx: *mut P2eCtx, map: *const P2eWireMap, cols: *const u64, ld: usize, aux: *const u64,
        ld_aux: usize, ux: *const c_void, ux_u32: i32, ld_ux: usize, gate: *const u64, ld_gate: usize, wires: *mut u64,
        wire_stride: usize, n: usize) -> i64;
    pub fn p2e_gate_internal_batch(ctx: *mut P2eCtx, program: i32, aux: *const u64, ld_aux: usize, gate: *mut u64,
        ld_gate: usize, n: usize) -> i64;
    pub fn p2e_gate_internal_num_cols(program: i32) -> i64;

    // ---- column maps (host only)
    pub fn p2e_schedule_describe(program: i32, out: *mut P2eGenDesc, cap: usize) -> i64;
    pub fn p2e_schedule_num_cols(program: i32) -> i64;
    pub fn p2e_schedule_wiring(program: i32, out: *mut P2eGenWiring, cap: usize) -> i64;
    pub fn p2e_wiring_const(id: u32, out32: *mut u8) -> i32;
    pub fn p2e_compact_layout(program: i32, col_map: *mut u32, cap: usize, num_narrow: *mut u32, num_wide: *mut u32) -> i64;

    // ---- layout helpers
    pub fn p2e_columns_to_rows(ctx: *mut P2eCtx, cols: *const u64, ld: usize, n: usize, ncols: usize, rows: *mut u64,
        row_ld: usize) -> i64;
    pub fn p2e_columns_compact(ctx: *mut P2eCtx, program: i32, cols: *const u64, ld: usize, n: usize, narrow: *mut u32,
        ld_narrow: usize, wide: *mut u64, ld_wide: usize, err: *mut u8) -> i64;
    pub fn p2e_compact_to_rows(ctx: *mut P2eCtx, program: i32, narrow: *const u32, ld_narrow: usize, wide: *const u64,
        ld_wide: usize, n: usize, rows_narrow: *mut u32, row_ld_narrow: usize, rows_wide: *mut u64, row_ld_wide: usize) -> i64;

    // ---- synthetic inputs (host only): valid signatures per curve/ecdsa.rs:25-40
    pub fn p2e_synth_signatures(seed: u64, first: usize, n: usize, msg32: *mut u8, r32: *mut u8, s32: *mut u8, pkx32: *mut u8,
        pky32: *mut u8) -> i32;
    pub fn p2e_synth_signatures_curve(curve: i32, seed: u64, first: usize, n: usize, msg32: *mut u8, r32: *mut u8, s32: *mut u8,
        pkx32: *mut u8, pky32: *mut u8) -> i32;

    // ---- curve programs (rank 4): curve_scalar_mul_windowed / curve_scalar_mul on either curve, verify_p256_message_circuit
    pub fn p2e_curve_program_create(ctx: *mut P2eCtx, kind: i32, curve: i32, blind_x32: *const u8, blind_y32: *const u8,
        out: *mut *mut P2eCurveProgram) -> i32;
    pub fn p2e_curve_program_destroy(ctx: *mut P2eCtx, prog: *mut P2eCurveProgram);
    pub fn p2e_curve_program_num_cols(prog: *const P2eCurveProgram) -> i64;
    pub fn p2e_curve_program_num_aux_cols(prog: *const P2eCurveProgram) -> i64;
    pub fn p2e_curve_program_scratch_bytes(prog: *const P2eCurveProgram, n: usize) -> usize;
    pub fn p2e_curve_program_describe(prog: *const P2eCurveProgram, out: *mut P2eGenDesc, cap: usize) -> i64;
    pub fn p2e_curve_program_wiring(prog: *const P2eCurveProgram, out: *mut P2eGenWiring, cap: usize) -> i64;
    pub fn p2e_curve_program_aux_describe(prog: *const P2eCurveProgram, out: *mut P2eAuxDesc, cap: usize) -> i64;
    pub fn p2e_curve_program_const(prog: *const P2eCurveProgram, id: u32, out32: *mut u8) -> i32;
    pub fn p2e_curve_program_num_gate_cols(prog: *const P2eCurveProgram) -> i64;
    pub fn p2e_curve_program_num_ux_cols(prog: *const P2eCurveProgram) -> i64;
    pub fn p2e_curve_program_ux_describe(prog: *const P2eCurveProgram, out: *mut P2eUxDesc, cap: usize) -> i64;
    pub fn p2e_curve_program_aux_witness_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, msg32: *const u8, r32: *const u8,
        s32: *const u8, pkx32: *const u8, pky32: *const u8, cols: *const u64, ld: usize, aux: *mut u64, ld_aux: usize, n: usize,
        err: *mut u8) -> i64;
    pub fn p2e_curve_program_gate_internal_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, aux: *const u64, ld_aux: usize,
        gate: *mut u64, ld_gate: usize, n: usize) -> i64;
    pub fn p2e_curve_program_ux_witness_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, msg32: *const u8, r32: *const u8,
        s32: *const u8, pkx32: *const u8, pky32: *const u8, cols: *const u64, ld: usize, aux: *const u64, ld_aux: usize,
        ux: *mut c_void, ux_u32: i32, ld_ux: usize, n: usize, err: *mut u8) -> i64;
    pub fn p2e_curve_program_wire_map_create(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, entries: *const P2eWireMapEntry,
        count: usize, num_wires: u32, degree: u32, out: *mut *mut P2eWireMap) -> i32;
    pub fn p2e_p256_verify_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, msg32: *const u8, r32: *const u8, s32: *const u8,
        pkx32: *const u8, pky32: *const u8, n: usize, err: *mut u8, valid: *mut u8) -> i64;
    pub fn p2e_curve_mul_witness_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, px32: *const u8, py32: *const u8,
        k32: *const u8, cols: *mut u64, n: usize, ld: usize, err: *mut u8, valid: *mut u8) -> i64;
    pub fn p2e_p256_verify_witness_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, msg32: *const u8, r32: *const u8,
        s32: *const u8, pkx32: *const u8, pky32: *const u8, cols: *mut u64, n: usize, ld: usize, err: *mut u8,
        valid: *mut u8) -> i64;
    // the same two fills into the compact container (u32 narrow + u64 wide matrices), and its column map
    pub fn p2e_curve_mul_witness_compact_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, px32: *const u8, py32: *const u8,
        k32: *const u8, narrow: *mut u32, ld_narrow: usize, wide: *mut u64, ld_wide: usize, n: usize, err: *mut u8,
        valid: *mut u8) -> i64;
    pub fn p2e_p256_verify_witness_compact_batch(ctx: *mut P2eCtx, prog: *const P2eCurveProgram, msg32: *const u8, r32: *const u8,
        s32: *const u8, pkx32: *const u8, pky32: *const u8, narrow: *mut u32, ld_narrow: usize, wide: *mut u64, ld_wide: usize,
        n: usize, err: *mut u8, valid: *mut u8) -> i64;
    pub fn p2e_curve_program_compact_layout(prog: *const P2eCurveProgram, col_map: *mut u32, cap: usize, num_narrow: *mut u32,
        num_wide: *mut u32) -> i64;
}
