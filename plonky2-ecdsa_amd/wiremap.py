"""Wire maps for p2e_assemble_wires (include/p2e.h).

The real map belongs to the Rust side: it is read off the targets the gadgets returned when the circuit was built
(INTEGRATION.md section 3).  plonky2 is not available offline, so what lives here is a SYNTHETIC placement with the
shape of ``CircuitConfig::standard_ecc_config()`` (136 wires): every MulNonnativeGate / CheckSumGate pair on its two
rows with the reference's wire layout (gates/mul_nonnative.rs:41-59, 384-390; the x / y wires repeat the operand
limbs and the CheckSumGate's a wires repeat check_sum: copy constraints, gadgets/nonnative.rs:406-445), every other
value packed row-major into the rows behind them the way arithmetic gates fill a row wire by wire.  It is good for
exercising and timing the scatter; it is not plonky2's placement."""
from __future__ import annotations

import numpy as np

from . import (PROGRAM_VERIFY, SRC_AUX, aux_num_cols, schedule_describe, schedule_wiring, ux_describe, ux_num_cols)

WIRE_SRC_COLS, WIRE_SRC_AUX, WIRE_SRC_UX, WIRE_SRC_GATE = 0x00000000, 0x40000000, 0x80000000, 0xC0000000


def synthetic_wire_map(program: int = PROGRAM_VERIFY, num_wires: int = 136, with_aux: bool = True, with_ux: bool = True,
                       with_gate: bool = False):
    """(src uint32[], dst uint32[], num_wires, degree): dst = wire * degree + row."""
    gens = schedule_describe(program)
    wiring = schedule_wiring(program)
    place = []                                   # (src, row, wire)
    row = 0
    packed = []                                  # sources that go row-major behind the gate rows

    def matrix_src(code, k):
        """library column of limb k of an operand, or None (inputs and constants are set by the host)"""
        if code & 0xC0000000:
            return None
        if code & SRC_AUX:
            return WIRE_SRC_AUX | ((code & ~SRC_AUX) + k)
        return WIRE_SRC_COLS | (code + k)

    for (kind, _field, c0, nc, _label), (ops, _rc) in zip(gens, wiring):
        if kind == "mul":
            for oi, (code, nl) in enumerate(ops):            # x wires 0..8, y wires 9..17
                for k in range(nl):
                    s = matrix_src(code, k)
                    if s is not None:
                        place.append((s, row, 9 * oi + k))
            for k in range(9):
                place.append((WIRE_SRC_COLS | (c0 + k), row, 18 + k))            # r
                place.append((WIRE_SRC_COLS | (c0 + 9 + k), row, 27 + k))        # q
            for k in range(17):
                place.append((WIRE_SRC_COLS | (c0 + 18 + k), row, 36 + k))       # check_sum
                place.append((WIRE_SRC_COLS | (c0 + 18 + k), row + 1, k))        # CheckSumGate a(i)
            for k in range(16):
                place.append((WIRE_SRC_COLS | (c0 + 35 + k), row + 1, 17 + k))   # b
            row += 2
        else:
            packed.extend(WIRE_SRC_COLS | (c0 + k) for k in range(nc))
    if with_aux:
        packed.extend(WIRE_SRC_AUX | c for c in range(aux_num_cols(program)))
    if with_ux:
        packed.extend(WIRE_SRC_UX | c for c in range(ux_num_cols(program)))
    if with_gate:
        from . import VERIFY_GATE_COLS, GLV_MUL_GATE_COLS
        packed.extend(WIRE_SRC_GATE | c for c in range(VERIFY_GATE_COLS if program == PROGRAM_VERIFY else GLV_MUL_GATE_COLS))
    first_packed_row = row
    rows = first_packed_row + -(-len(packed) // num_wires)
    degree = 1 << max(1, (rows - 1).bit_length())
    src = np.empty(len(place) + len(packed), dtype=np.uint32)
    dst = np.empty_like(src)
    for j, (s, r, w) in enumerate(place):
        src[j] = s
        dst[j] = w * degree + r
    k = np.arange(len(packed), dtype=np.int64)
    src[len(place):] = np.array(packed, dtype=np.uint32)
    dst[len(place):] = ((k % num_wires) * degree + first_packed_row + k // num_wires).astype(np.uint32)
    assert len(np.unique(dst)) == len(dst)
    return src, dst, num_wires, degree
