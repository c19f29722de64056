"""plonky2-ecdsa_amd: MI355X-native batch witness generation for plonky2's secp256k1 ECDSA gadget.

Python host side of the C ABI in ``include/p2e.h`` (``libp2e_hip.so``, hand-written HIP for gfx950).
The names mirror the reference crate's generator / gadget interface for this path
(Weobe/plonky2-ecdsa: ``MulNonnativeGenerator``, ``NonNative{Addition,Subtraction,MultipleAdds,
Inverse}Generator``, ``GLVDecompositionGenerator``, ``glv_mul``, ``verify_secp256k1_message_circuit``),
with batches instead of single targets:

* Goldilocks columns are ``(num_cols, n)`` uint64 arrays / tensors, column-major over the batch.
* 256-bit inputs are ``(n, 32)`` uint8 little-endian.

There is NO CPU fallback: without the HIP extension and a GPU every compute call raises.
torch is used only for device memory / streams / torch.distributed plumbing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels sharing a queue
# run in order.  The fused entry points pipeline over three streams per context; with 8 queues they never
# collide with the caller's own streams (+5 % on the 2^16 verify batch).  Read once, when the HIP runtime
# starts: this only has an effect if nothing initialised HIP before this import.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# P2E_LIB: an alternative build of the same library (A/B experiments under tools/); never set in production
LIB_PATH = os.environ.get("P2E_LIB") or os.path.join(_HERE, "libp2e_hip.so")

FIELD_BASE = 0    # plonky2 Secp256K1Base
FIELD_SCALAR = 1  # plonky2 Secp256K1Scalar
FIELD_P256_BASE, FIELD_P256_SCALAR = 2, 3   # the crate's P256Base / P256Scalar (single-generator entry points)
ERR_LIMB_RANGE, ERR_VALUE_GE_2_256, ERR_INVERSE_OF_ZERO, ERR_CARRY_RANGE, ERR_QUOTIENT_RANGE = 1, 2, 4, 8, 16
ERR_DIVISION_BY_ZERO = 32
CTX_HOST_POINTERS, CTX_ASYNC, CTX_PHASE_TIMING = 1, 2, 4
VERIFY_COLS = 82615
GLV_MUL_COLS = 65243
VERIFY_AUX_COLS = 8959      # built-in-generator columns (include/p2e.h p2e_aux_witness_batch)
GLV_MUL_AUX_COLS = 4738
VERIFY_UX_COLS = 249385     # constraint-block (U29 gate) columns (include/p2e.h p2e_ux_witness_batch)
GLV_MUL_UX_COLS = 194361
VERIFY_GATE_COLS = 10703    # gate-internal values of the built-in gates (include/p2e.h p2e_gate_internal_batch)
GLV_MUL_GATE_COLS = 5621
PROGRAM_VERIFY, PROGRAM_GLV_MUL = 0, 1
# curve programs (include/p2e.h P2E_CURVE_* / P2E_CP_*)
CURVE_SECP256K1, CURVE_P256 = 0, 1
CP_WINDOWED_MUL, CP_SCALAR_MUL, CP_VERIFY = 1, 2, 3
WINDOWED_MUL_COLS, SCALAR_MUL_COLS, P256_VERIFY_COLS = 98185, 139354, 115557

# every symbol include/p2e.h declares
EXPORTS = (
    "p2e_ctx_create", "p2e_ctx_destroy", "p2e_sync", "p2e_last_error", "p2e_scratch_bytes", "p2e_last_phase_ms",
    "p2e_segments_describe", "p2e_segment_stream_wait", "p2e_segment_sync",
    "p2e_mul_witness_batch", "p2e_checksum_witness_batch", "p2e_add_witness_batch", "p2e_sub_witness_batch",
    "p2e_add_many_witness_batch", "p2e_inv_witness_batch", "p2e_glv_decompose_batch", "p2e_limb_split",
    "p2e_limb_pack", "p2e_ecdsa_verify_witness_batch", "p2e_glv_mul_witness_batch", "p2e_columns_to_rows",
    "p2e_schedule_describe", "p2e_schedule_wiring", "p2e_wiring_const", "p2e_ux_witness_batch", "p2e_ux_describe", "p2e_ux_num_cols",
    "p2e_wire_map_create", "p2e_wire_map_destroy", "p2e_assemble_wires", "p2e_gate_internal_batch", "p2e_gate_internal_num_cols",
    "p2e_schedule_num_cols", "p2e_synth_signatures", "p2e_aux_witness_batch", "p2e_aux_describe", "p2e_aux_num_cols",
    "p2e_compact_layout", "p2e_columns_compact", "p2e_ecdsa_verify_witness_compact_batch",
    "p2e_glv_mul_witness_compact_batch", "p2e_aux_witness_compact_batch", "p2e_compact_to_rows", "p2e_ecdsa_verify_batch", "p2e_biguint_div_rem_batch",
    "p2e_curve_program_create", "p2e_curve_program_destroy", "p2e_curve_program_num_cols", "p2e_curve_program_num_aux_cols",
    "p2e_curve_program_scratch_bytes", "p2e_curve_program_describe", "p2e_curve_program_wiring", "p2e_curve_program_aux_describe",
    "p2e_curve_program_const", "p2e_curve_mul_witness_batch", "p2e_p256_verify_witness_batch", "p2e_synth_signatures_curve",
    "p2e_curve_program_num_gate_cols", "p2e_curve_program_num_ux_cols", "p2e_curve_program_ux_describe",
    "p2e_curve_program_aux_witness_batch", "p2e_curve_program_gate_internal_batch", "p2e_curve_program_ux_witness_batch",
    "p2e_curve_program_wire_map_create", "p2e_p256_verify_batch",
    "p2e_curve_mul_witness_compact_batch", "p2e_p256_verify_witness_compact_batch", "p2e_curve_program_compact_layout",
)


class P2EError(RuntimeError):
    pass


def build(verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU).  csrc/p2e_hip.hip is ONE source
    compiled as four translation units side by side (-DP2E_PART=0..3, see the top of the file) and linked into
    libp2e_hip.so.  ``extra_flags`` / ``out``: experimental builds for tools/ (never the product library)."""
    from concurrent.futures import ThreadPoolExecutor
    src = os.path.join(_HERE, "csrc", "p2e_hip.hip")
    lib_path = out or LIB_PATH
    deps = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    deps.append(os.path.join(_ROOT, "include", "p2e.h"))
    if not extra_flags and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in deps):
        return lib_path
    objdir = os.path.join(_HERE, "build", os.path.basename(lib_path))
    os.makedirs(objdir, exist_ok=True)
    common = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", *extra_flags]

    def compile_part(part):
        obj = os.path.join(objdir, f"part{part}.o")
        cmd = common + [f"-DP2E_PART={part}", "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_part, range(4)))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fopenmp", "-o", lib_path, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


_lib = None


class _GenDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("field", C.c_int32), ("first_col", C.c_uint32), ("num_cols", C.c_uint32),
                ("label", C.c_char * 48)]


def lib():
    """Load libp2e_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P2EError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
        # torch ships its own libamdhip64 / libhsa-runtime64.  Import it FIRST so that libp2e_hip.so resolves
        # its libamdhip64.so.7 dependency to the runtime torch already loaded: device pointers and streams
        # are only meaningful inside one HIP runtime, and a second HSA runtime in the process finds no GPU
        # ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        experimental = bool(os.environ.get("P2E_LIB"))   # an A/B build under tools/ may predate newer entry points
        for name in EXPORTS:
            if experimental and not hasattr(_lib, name):
                continue
            getattr(_lib, name)  # AttributeError if a declared symbol is not exported
        _lib.p2e_last_error.restype = C.c_char_p
        _lib.p2e_scratch_bytes.restype = C.c_size_t
        _lib.p2e_scratch_bytes.argtypes = [C.c_int, C.c_size_t]
        for name in EXPORTS:
            if experimental and not hasattr(_lib, name):
                continue
            if name.endswith("_batch") or name in ("p2e_limb_split", "p2e_limb_pack", "p2e_columns_to_rows", "p2e_schedule_describe",
                                                   "p2e_schedule_num_cols", "p2e_aux_describe", "p2e_aux_num_cols", "p2e_compact_layout",
                                                   "p2e_columns_compact", "p2e_compact_to_rows", "p2e_curve_program_num_cols",
                                                   "p2e_curve_program_num_aux_cols", "p2e_curve_program_describe",
                                                   "p2e_curve_program_wiring", "p2e_curve_program_aux_describe",
                                                   "p2e_curve_program_num_gate_cols", "p2e_curve_program_num_ux_cols",
                                                   "p2e_curve_program_ux_describe", "p2e_curve_program_compact_layout"):
                getattr(_lib, name).restype = C.c_long
        _lib.p2e_curve_program_scratch_bytes.restype = C.c_size_t
    return _lib


def _ld(a):
    """Row stride, in elements, of a 2-D column matrix (numpy or torch): what the C ABI calls `ld`.  A view of a
    wider matrix (e.g. the padded output of ecdsa_verify_witness_batch) carries its stride with it; the inner
    (batch) dimension must be dense."""
    if isinstance(a, np.ndarray):
        if a.ndim != 2 or (a.shape[1] > 1 and a.strides[1] != a.itemsize):
            raise P2EError("column matrices must be 2-D with a dense batch dimension")
        return a.strides[0] // a.itemsize if a.shape[0] > 1 else max(a.shape[1], a.strides[0] // a.itemsize)
    if a.dim() != 2 or (a.shape[1] > 1 and a.stride(1) != 1):
        raise P2EError("column matrices must be 2-D with a dense batch dimension")
    return a.stride(0) if a.shape[0] > 1 else max(a.shape[1], a.stride(0))


def _dense(n, *arrays):
    """the single-generator entry points share ONE ld between inputs and outputs: inputs must be dense (k, n)"""
    for a in arrays:
        if _ld(a.reshape(-1, a.shape[-1]) if a.ndim == 3 else a) != n:
            raise P2EError("pass contiguous (k, n) limb-column arrays")


def _ptr(a):
    """device pointer of a torch tensor / host pointer of a numpy array / None"""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())


def schedule_describe(program: int = PROGRAM_VERIFY):
    """Column map: list of (kind, field, first_col, num_cols, label) in generator registration order."""
    L = lib()
    n = L.p2e_schedule_describe(C.c_int(program), None, C.c_size_t(0))
    arr = (_GenDesc * n)()
    L.p2e_schedule_describe(C.c_int(program), arr, C.c_size_t(n))
    kinds = ("add", "sub", "add_many", "mul", "inv", "glv")
    return [(kinds[d.kind], d.field, d.first_col, d.num_cols, d.label.decode()) for d in arr]


SRC_AUX, SRC_INPUT, SRC_CONST = 0x20000000, 0x40000000, 0x80000000
INPUT_SLOTS = ("pky", "pkx", "msg", "r", "s")


class _GenWiring(C.Structure):
    _fields_ = [("num_operands", C.c_int32), ("src", C.c_uint32 * 4), ("num_limbs", C.c_uint8 * 4), ("range_check", C.c_int32)]


def schedule_wiring(program: int = PROGRAM_VERIFY):
    """Operand wiring per generator (include/p2e.h p2e_schedule_wiring): list of ([(src, num_limbs), ...], range_check)
    in the order of schedule_describe."""
    L = lib()
    L.p2e_schedule_wiring.restype = C.c_long
    n = L.p2e_schedule_wiring(C.c_int(program), None, C.c_size_t(0))
    arr = (_GenWiring * n)()
    L.p2e_schedule_wiring(C.c_int(program), arr, C.c_size_t(n))
    return [([(int(w.src[k]), int(w.num_limbs[k])) for k in range(w.num_operands)], bool(w.range_check)) for w in arr]


def wiring_const(const_id: int):
    """(value, num_limbs) of circuit constant `const_id` (SRC_CONST | const_id in schedule_wiring)."""
    buf = (C.c_uint8 * 32)()
    nl = lib().p2e_wiring_const(C.c_uint32(const_id), buf)
    if nl < 0:
        raise P2EError("unknown constant id")
    return int.from_bytes(bytes(buf), "little"), int(nl)


class _UxDesc(C.Structure):
    _fields_ = [("first_col", C.c_uint32), ("num_cols", C.c_uint32)]


def ux_describe(program: int = PROGRAM_VERIFY):
    """Per generator (order of schedule_describe): (first_col, num_cols) of its block in the constraint-block matrix."""
    L = lib()
    L.p2e_ux_describe.restype = C.c_long
    n = L.p2e_ux_describe(C.c_int(program), None, C.c_size_t(0))
    arr = (_UxDesc * n)()
    L.p2e_ux_describe(C.c_int(program), arr, C.c_size_t(n))
    return [(int(d.first_col), int(d.num_cols)) for d in arr]


def ux_num_cols(program: int = PROGRAM_VERIFY) -> int:
    L = lib()
    L.p2e_ux_num_cols.restype = C.c_long
    return int(L.p2e_ux_num_cols(C.c_int(program)))


def schedule_num_cols(program: int = PROGRAM_VERIFY) -> int:
    return int(lib().p2e_schedule_num_cols(C.c_int(program)))


class _AuxDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("first_col", C.c_uint32), ("num_cols", C.c_uint32), ("label", C.c_char * 48)]


def aux_describe(program: int = PROGRAM_VERIFY):
    """Column map of the built-in-generator columns: (kind, first_col, num_cols, label) in gadget order."""
    L = lib()
    n = L.p2e_aux_describe(C.c_int(program), None, C.c_size_t(0))
    arr = (_AuxDesc * n)()
    L.p2e_aux_describe(C.c_int(program), arr, C.c_size_t(n))
    kinds = ("split4", "split2", "fixed_base_window", "msm_digit", "conditional_neg")
    return [(kinds[d.kind], d.first_col, d.num_cols, d.label.decode()) for d in arr]


COMPACT_WIDE = 0x80000000


def compact_layout(program: int = PROGRAM_VERIFY):
    """(col_map, num_narrow, num_wide) of the compact transfer container (include/p2e.h p2e_columns_compact):
    col_map[c] = index of witness column c in the u32 narrow matrix, or COMPACT_WIDE | index in the u64 wide one."""
    L = lib()
    ncols = schedule_num_cols(program)
    m = np.zeros(ncols, dtype=np.uint32)
    nn, nw = C.c_uint32(), C.c_uint32()
    L.p2e_compact_layout(C.c_int(program), _ptr(m), C.c_size_t(ncols), C.byref(nn), C.byref(nw))
    return m, int(nn.value), int(nw.value)


def compact_expand(program, narrow, wide):
    """Host-side inverse of Context.columns_compact (numpy): the (num_cols, n) u64 matrix."""
    m, nn, nw = compact_layout(program)
    out = np.empty((len(m), narrow.shape[1]), dtype=np.uint64)
    is_wide = (m & COMPACT_WIDE) != 0
    out[~is_wide] = np.asarray(narrow)[m[~is_wide]]
    out[is_wide] = np.asarray(wide).view(np.uint64)[m[is_wide] & 0x7FFFFFFF]
    return out


def aux_num_cols(program: int = PROGRAM_VERIFY) -> int:
    return int(lib().p2e_aux_num_cols(C.c_int(program)))


def synth_signatures(seed: int, n: int, first: int = 0):
    """Valid secp256k1 signatures (msg, r, s, pk.x, pk.y) as five (n, 32) uint8 arrays (host)."""
    out = [np.zeros((n, 32), dtype=np.uint8) for _ in range(5)]
    rc = lib().p2e_synth_signatures(C.c_uint64(seed), C.c_size_t(first), C.c_size_t(n), *[_ptr(a) for a in out])
    if rc:
        raise P2EError("p2e_synth_signatures failed")
    return out


def synth_signatures_curve(curve: int, seed: int, n: int, first: int = 0):
    """Valid signatures on a curve of the crate (CURVE_*): (msg, r, s, pk.x, pk.y) as five (n, 32) uint8 arrays (host)."""
    out = [np.zeros((n, 32), dtype=np.uint8) for _ in range(5)]
    rc = lib().p2e_synth_signatures_curve(C.c_int(curve), C.c_uint64(seed), C.c_size_t(first), C.c_size_t(n), *[_ptr(a) for a in out])
    if rc:
        raise P2EError("p2e_synth_signatures_curve failed")
    return out


class CurveProgram:
    """One built circuit of curve_scalar_mul_windowed / curve_scalar_mul / verify_p256_message_circuit
    (gadgets/curve_windowed_mul.rs:131-173, gadgets/curve.rs:245-285, gadgets/ecdsa.rs:55-78) on CURVE_SECP256K1 or
    CURVE_P256.  ``blind`` = the point the gadget draws with rand() while the circuit is built, as (x, y) ints."""

    def __init__(self, ctx, kind: int, curve: int, blind):
        self._ctx, self.kind, self.curve = ctx, kind, curve
        bx = np.frombuffer(int(blind[0]).to_bytes(32, "little"), np.uint8).copy()
        by = np.frombuffer(int(blind[1]).to_bytes(32, "little"), np.uint8).copy()
        h = C.c_void_p()
        rc = ctx._L.p2e_curve_program_create(ctx._h, C.c_int(kind), C.c_int(curve), _ptr(bx), _ptr(by), C.byref(h))
        if rc != 0:
            raise P2EError(f"p2e_curve_program_create failed ({rc}): {ctx._L.p2e_last_error().decode()}")
        self._h = h
        self.num_cols = int(ctx._L.p2e_curve_program_num_cols(h))
        self.num_aux_cols = int(ctx._L.p2e_curve_program_num_aux_cols(h))
        self.num_gate_cols = int(ctx._L.p2e_curve_program_num_gate_cols(h))
        self.num_ux_cols = int(ctx._L.p2e_curve_program_num_ux_cols(h))

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._ctx._L.p2e_curve_program_destroy(self._ctx._h, self._h)
        self._h = None

    __del__ = close

    def scratch_bytes(self, n):
        return int(self._ctx._L.p2e_curve_program_scratch_bytes(self._h, C.c_size_t(n)))

    def describe(self):
        """(kind, field, first_col, num_cols, label) per generator, registration order (as schedule_describe)."""
        L = self._ctx._L
        n = L.p2e_curve_program_describe(self._h, None, C.c_size_t(0))
        arr = (_GenDesc * n)()
        L.p2e_curve_program_describe(self._h, arr, C.c_size_t(n))
        kinds = ("add", "sub", "add_many", "mul", "inv", "glv")
        return [(kinds[d.kind], d.field, d.first_col, d.num_cols, d.label.decode()) for d in arr]

    def wiring(self):
        """([(src, num_limbs), ...], range_check) per generator (as schedule_wiring); constants through const()."""
        L = self._ctx._L
        n = L.p2e_curve_program_wiring(self._h, None, C.c_size_t(0))
        arr = (_GenWiring * n)()
        L.p2e_curve_program_wiring(self._h, arr, C.c_size_t(n))
        return [([(int(w.src[k]), int(w.num_limbs[k])) for k in range(w.num_operands)], bool(w.range_check)) for w in arr]

    def aux_describe(self):
        """(kind code, first_col, num_cols, label) of the built-in-generator targets the wiring refers to."""
        L = self._ctx._L
        n = L.p2e_curve_program_aux_describe(self._h, None, C.c_size_t(0))
        arr = (_AuxDesc * n)()
        L.p2e_curve_program_aux_describe(self._h, arr, C.c_size_t(n))
        return [(int(d.kind), d.first_col, d.num_cols, d.label.decode()) for d in arr]

    def const(self, const_id: int) -> int:
        buf = (C.c_uint8 * 32)()
        if self._ctx._L.p2e_curve_program_const(self._h, C.c_uint32(const_id), buf) != 0:
            raise P2EError("unknown constant id")
        return int.from_bytes(bytes(buf), "little")

    def _out(self, n, cols, err, valid, ld):
        ctx = self._ctx
        if cols is None:
            ld = n + 16 if (n >= 4096 and n & (n - 1) == 0) else n
            full = ctx._cols(self.num_cols, ld)
            cols = full[:, :n] if ld != n else full
        err = err if err is not None else ctx._vec(n, np.uint8)
        valid = valid if valid is not None else ctx._vec(n, np.uint8)
        return cols, err, valid, ld if ld is not None else _ld(cols)

    def mul_witness_batch(self, px, py, k, cols=None, err=None, valid=None, ld=None):
        """(num_cols, n) columns of every generator the gadget registers for n (point, scalar) pairs."""
        ctx = self._ctx
        n = ctx._shape(px)[0]
        cols, err, valid, ld = self._out(n, cols, err, valid, ld)
        bad = ctx._check(ctx._L.p2e_curve_mul_witness_batch(ctx._h, self._h, _ptr(px), _ptr(py), _ptr(k), _ptr(cols), C.c_size_t(n),
                                                            C.c_size_t(ld), _ptr(err), _ptr(valid)))
        return cols, err, valid, bad

    # ---- the compact container (include/p2e.h p2e_curve_program_compact_layout) -----------------------------------
    def compact_layout(self):
        """(col_map, num_narrow, num_wide): col_map[c] = row of witness column c in the u32 narrow matrix, or
        COMPACT_WIDE | row in the u64 wide one."""
        L = self._ctx._L
        m = np.zeros(self.num_cols, dtype=np.uint32)
        nn, nw = C.c_uint32(), C.c_uint32()
        L.p2e_curve_program_compact_layout(self._h, _ptr(m), C.c_size_t(self.num_cols), C.byref(nn), C.byref(nw))
        return m, int(nn.value), int(nw.value)

    def compact_expand(self, narrow, wide):
        """host-side inverse (numpy): the (num_cols, n) u64 matrix of a compact container of this program"""
        m, _nn, _nw = self.compact_layout()
        out = np.empty((len(m), narrow.shape[1]), dtype=np.uint64)
        is_wide = (m & COMPACT_WIDE) != 0
        out[~is_wide] = np.asarray(narrow).view(np.uint32)[m[~is_wide]]
        out[is_wide] = np.asarray(wide).view(np.uint64)[m[is_wide] & 0x7FFFFFFF]
        return out

    def _compact_out(self, n, narrow, wide, err, valid, ld_narrow, ld_wide):
        ctx = self._ctx
        _m, nn, nw = self.compact_layout()
        if narrow is None or wide is None:
            ldp = n + 16 if (not ctx.host_pointers and n >= 4096 and n & (n - 1) == 0) else n
            if ctx.host_pointers:
                narrow, wide = np.zeros((nn, ldp), dtype=np.uint32), np.zeros((nw, ldp), dtype=np.uint64)
            else:
                import torch
                dev = f"cuda:{ctx.device}"
                narrow = torch.empty((nn, ldp), dtype=torch.int32, device=dev)
                wide = torch.empty((nw, ldp), dtype=torch.int64, device=dev)
            narrow, wide = narrow[:, :n], wide[:, :n]
        err = err if err is not None else ctx._vec(n, np.uint8)
        valid = valid if valid is not None else ctx._vec(n, np.uint8)
        return narrow, wide, err, valid, ld_narrow or _ld(narrow), ld_wide or _ld(wide)

    def mul_witness_compact_batch(self, px, py, k, narrow=None, wide=None, err=None, valid=None, ld_narrow=None, ld_wide=None):
        """mul_witness_batch writing the compact container: (narrow u32, wide u64, err, valid, bad)"""
        ctx = self._ctx
        n = ctx._shape(px)[0]
        narrow, wide, err, valid, ldn, ldw = self._compact_out(n, narrow, wide, err, valid, ld_narrow, ld_wide)
        ctx._L.p2e_curve_mul_witness_compact_batch.restype = C.c_long
        bad = ctx._check(ctx._L.p2e_curve_mul_witness_compact_batch(ctx._h, self._h, _ptr(px), _ptr(py), _ptr(k), _ptr(narrow), C.c_size_t(ldn),
                                                                    _ptr(wide), C.c_size_t(ldw), C.c_size_t(n), _ptr(err), _ptr(valid)))
        return narrow, wide, err, valid, bad

    def verify_witness_compact_batch(self, msg, r, s, pkx, pky, narrow=None, wide=None, err=None, valid=None, ld_narrow=None, ld_wide=None):
        """verify_witness_batch writing the compact container: (narrow u32, wide u64, err, valid, bad)"""
        ctx = self._ctx
        n = ctx._shape(msg)[0]
        narrow, wide, err, valid, ldn, ldw = self._compact_out(n, narrow, wide, err, valid, ld_narrow, ld_wide)
        ctx._L.p2e_p256_verify_witness_compact_batch.restype = C.c_long
        bad = ctx._check(ctx._L.p2e_p256_verify_witness_compact_batch(ctx._h, self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky),
                                                                      _ptr(narrow), C.c_size_t(ldn), _ptr(wide), C.c_size_t(ldw), C.c_size_t(n),
                                                                      _ptr(err), _ptr(valid)))
        return narrow, wide, err, valid, bad

    def _inputs(self, inputs):
        """(msg, r, s, pkx, pky) pointers from the program's input tuple: (px, py, k) or (msg, r, s, pkx, pky)"""
        if len(inputs) == 3:
            px, py, k = inputs
            return k, None, None, px, py
        return tuple(inputs)

    def ux_describe(self):
        """(first_col, num_cols) of every generator's constraint block (order of describe())."""
        L = self._ctx._L
        n = L.p2e_curve_program_ux_describe(self._h, None, C.c_size_t(0))
        arr = (_UxDesc * n)()
        L.p2e_curve_program_ux_describe(self._h, arr, C.c_size_t(n))
        return [(int(d.first_col), int(d.num_cols)) for d in arr]

    def aux_witness_batch(self, inputs, cols, n=None, ld=None, aux=None, err=None):
        """built-in-generator targets of the program's circuit from its finished witness matrix: (num_aux_cols, n)"""
        ctx = self._ctx
        msg, r, s, px, py = self._inputs(inputs)
        n = n if n is not None else ctx._shape(px)[0]
        ld = ld if ld is not None else _ld(cols)
        aux = aux if aux is not None else ctx._cols(self.num_aux_cols, n)
        err = err if err is not None else ctx._vec(n, np.uint8)
        bad = ctx._check(ctx._L.p2e_curve_program_aux_witness_batch(ctx._h, self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(px), _ptr(py),
                                                                    _ptr(cols), C.c_size_t(ld), _ptr(aux), C.c_size_t(_ld(aux)),
                                                                    C.c_size_t(n), _ptr(err)))
        return aux, err, bad

    def gate_internal_batch(self, aux, n=None, gate=None):
        """gate-internal values of the built-in gates behind the windows, from the aux matrix: (num_gate_cols, n)"""
        ctx = self._ctx
        n = n if n is not None else aux.shape[1]
        gate = gate if gate is not None else ctx._cols(self.num_gate_cols, n)
        ctx._check(ctx._L.p2e_curve_program_gate_internal_batch(ctx._h, self._h, _ptr(aux), C.c_size_t(_ld(aux)), _ptr(gate),
                                                                C.c_size_t(_ld(gate)), C.c_size_t(n)))
        return gate

    def ux_witness_batch(self, inputs, cols, aux, n=None, ld=None, ux=None, err=None, u32=True):
        """constraint-block (U29 gate) values from the finished witness and aux matrices: (num_ux_cols, n) u32 / u64"""
        ctx = self._ctx
        msg, r, s, px, py = self._inputs(inputs)
        n = n if n is not None else ctx._shape(px)[0]
        ld = ld if ld is not None else _ld(cols)
        if ux is None:
            if ctx.host_pointers:
                ux = np.zeros((self.num_ux_cols, n), dtype=np.uint32 if u32 else np.uint64)
            else:
                import torch
                ux = torch.empty((self.num_ux_cols, n), dtype=torch.int32 if u32 else torch.int64, device=f"cuda:{ctx.device}")
        err = err if err is not None else ctx._vec(n, np.uint8)
        bad = ctx._check(ctx._L.p2e_curve_program_ux_witness_batch(ctx._h, self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(px), _ptr(py),
                                                                   _ptr(cols), C.c_size_t(ld), _ptr(aux), C.c_size_t(_ld(aux)), _ptr(ux),
                                                                   C.c_int(1 if u32 else 0), C.c_size_t(_ld(ux)), C.c_size_t(n), _ptr(err)))
        return ux, err, bad

    def verify_batch(self, msg, r, s, pkx, pky, err=None, valid=None):
        """the verifier circuit's verdict alone (no witness): (err, valid, flagged count)"""
        ctx = self._ctx
        n = ctx._shape(msg)[0]
        err = err if err is not None else ctx._vec(n, np.uint8)
        valid = valid if valid is not None else ctx._vec(n, np.uint8)
        bad = ctx._check(ctx._L.p2e_p256_verify_batch(ctx._h, self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky), C.c_size_t(n),
                                                      _ptr(err), _ptr(valid)))
        return err, valid, bad

    def wire_map(self, src, dst, num_wires, degree):
        """a column -> (wire, row) map over THIS program's four matrices, for Context.assemble_wires"""
        ctx = self._ctx
        src, dst = np.ascontiguousarray(src, np.uint32), np.ascontiguousarray(dst, np.uint32)
        ent = np.empty((len(src), 2), dtype=np.uint32)
        ent[:, 0], ent[:, 1] = src, dst
        h = C.c_void_p()
        rc = ctx._L.p2e_curve_program_wire_map_create(ctx._h, self._h, _ptr(ent), C.c_size_t(len(src)), C.c_uint32(num_wires),
                                                      C.c_uint32(degree), C.byref(h))
        if rc != 0:
            raise P2EError(f"p2e_curve_program_wire_map_create failed ({rc}): {ctx._L.p2e_last_error().decode()}")
        return _WireMap(ctx, h, -1, num_wires, degree, len(src))

    def verify_witness_batch(self, msg, r, s, pkx, pky, cols=None, err=None, valid=None, ld=None):
        """verify_p256_message_circuit: (115557, n) columns."""
        ctx = self._ctx
        n = ctx._shape(msg)[0]
        cols, err, valid, ld = self._out(n, cols, err, valid, ld)
        bad = ctx._check(ctx._L.p2e_p256_verify_witness_batch(ctx._h, self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky),
                                                              _ptr(cols), C.c_size_t(n), C.c_size_t(ld), _ptr(err), _ptr(valid)))
        return cols, err, valid, bad


class _WireMap:
    def __init__(self, ctx, h, program, num_wires, degree, count):
        self._ctx, self._h, self.program, self.num_wires, self.degree, self.count = ctx, h, program, num_wires, degree, count

    def close(self):
        if self._h and getattr(self._ctx, "_h", None):
            self._ctx._L.p2e_wire_map_destroy(self._ctx._h, self._h)
        self._h = None

    __del__ = close


class Context:
    """One p2e_ctx: one per (host thread, device).

    ``host_pointers=True`` makes every entry point take numpy arrays (staged through the library's own
    device buffers: convenient for tests, PCIe-inclusive, never benchmarked).  Otherwise arguments are
    torch CUDA tensors and the context runs on ``stream`` (default: torch's current stream).
    ``phase_timing=True`` (P2E_CTX_PHASE_TIMING) times every launch for last_phase_ms(); off by default, it costs 6 % at 2^13
    signatures per call."""

    def __init__(self, device: int = 0, host_pointers: bool = False, stream=None, asynchronous: bool = False,
                 phase_timing: bool = False):
        L = lib()
        self._L = L
        self.host_pointers = host_pointers
        flags = ((CTX_HOST_POINTERS if host_pointers else 0) | (CTX_ASYNC if asynchronous else 0)
                 | (CTX_PHASE_TIMING if phase_timing else 0))
        if stream is None and not host_pointers:
            import torch
            stream = torch.cuda.current_stream(device).cuda_stream
        h = C.c_void_p()
        rc = L.p2e_ctx_create(C.c_int(device), C.c_uint(flags), C.c_void_p(stream or 0), C.byref(h))
        if rc != 0:
            raise P2EError(f"p2e_ctx_create failed ({rc}): {L.p2e_last_error().decode()}")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.p2e_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise P2EError(f"p2e call failed ({rc}): {self._L.p2e_last_error().decode()}")
        return int(rc)

    def sync(self):
        return self._check(self._L.p2e_sync(self._h))

    def last_phase_ms(self):
        """include/p2e.h p2e_last_phase_ms: `expand*` = k_expand launches, `runs*` = k_expand_runs launches, `fbrun*` =
        k_expand_fb_run launches (per kind: launches, columns written per signature, summed HIP-event durations in ms).
        Durations are 0 unless the context was created with phase_timing=True (launches and columns are always there)."""
        buf = (C.c_float * 12)()
        self._L.p2e_last_phase_ms(self._h, buf, C.c_int(12))
        return dict(zip(("scalar", "expand_launches", "expand_cols", "expand", "total", "runs_launches", "runs_cols",
                         "runs", "fbrun_launches", "fbrun_cols", "fbrun"), [float(x) for x in buf]))

    # ---- column blocks of the last fused call, as they become final (include/p2e.h p2e_segments_describe) -------------
    def segments(self):
        """[(first_col, num_cols)] of the last fused call issued on this context, in issue order: the scalar phase's
        blocks, then one block per expansion launch.  Disjoint; together every column of the program."""
        self._L.p2e_segments_describe.restype = C.c_long
        k = self._L.p2e_segments_describe(self._h, None, C.c_size_t(0))
        buf = (C.c_uint32 * (2 * max(k, 1)))()
        self._L.p2e_segments_describe(self._h, buf, C.c_size_t(k))
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(k)]

    def segment_stream_wait(self, k: int, stream: int):
        """`stream` (a raw hipStream_t handle, e.g. torch.cuda.Stream().cuda_stream) waits until block k is final."""
        if self._L.p2e_segment_stream_wait(self._h, C.c_int(k), C.c_void_p(stream)) != 0:
            raise P2EError(self._L.p2e_last_error().decode())

    def segment_sync(self, k: int):
        """the host blocks until block k is final"""
        if self._L.p2e_segment_sync(self._h, C.c_int(k)) != 0:
            raise P2EError(self._L.p2e_last_error().decode())

    # ---- allocation helpers -------------------------------------------------------------------------
    def _cols(self, k, n):
        if self.host_pointers:
            return np.zeros((k, n), dtype=np.uint64)
        import torch
        return torch.empty((k, n), dtype=torch.int64, device=f"cuda:{self.device}")

    def _vec(self, n, dtype=np.uint64):
        if self.host_pointers:
            return np.zeros(n, dtype=dtype)
        import torch
        td = {np.uint64: torch.int64, np.uint8: torch.uint8}[dtype]
        return torch.empty(n, dtype=td, device=f"cuda:{self.device}")

    @staticmethod
    def _shape(a):
        return tuple(a.shape)

    # ---- single generators ---------------------------------------------------------------------------
    def mul_witness_batch(self, field, x, y):
        """MulNonnativeGenerator + CheckSumGenerator (gates/mul_nonnative.rs:249-324, :513-531)."""
        n = self._shape(x)[1]
        _dense(n, x, y)
        r, q, cs, b, err = self._cols(9, n), self._cols(9, n), self._cols(17, n), self._cols(16, n), self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_mul_witness_batch(self._h, C.c_int(field), _ptr(x), _ptr(y), _ptr(r), _ptr(q),
                                                        _ptr(cs), _ptr(b), C.c_size_t(n), C.c_size_t(n), _ptr(err)))
        return r, q, cs, b, err, bad

    def checksum_witness_batch(self, a):
        n = self._shape(a)[1]
        _dense(n, a)
        b, err = self._cols(16, n), self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_checksum_witness_batch(self._h, _ptr(a), _ptr(b), C.c_size_t(n), C.c_size_t(n), _ptr(err)))
        return b, err, bad

    def _binop(self, fn, field, a, b):
        n = self._shape(a)[1]
        _dense(n, a, b)
        out, ov, err = self._cols(9, n), self._vec(n), self._vec(n, np.uint8)
        bad = self._check(fn(self._h, C.c_int(field), _ptr(a), _ptr(b), _ptr(out), _ptr(ov), C.c_size_t(n),
                             C.c_size_t(n), _ptr(err)))
        return out, ov, err, bad

    def add_witness_batch(self, field, a, b):
        """NonNativeAdditionGenerator (gadgets/nonnative.rs:626-645)."""
        return self._binop(self._L.p2e_add_witness_batch, field, a, b)

    def sub_witness_batch(self, field, a, b):
        """NonNativeSubtractionGenerator (gadgets/nonnative.rs:792-810)."""
        return self._binop(self._L.p2e_sub_witness_batch, field, a, b)

    def add_many_witness_batch(self, field, summands):
        """NonNativeMultipleAddsGenerator (gadgets/nonnative.rs:696-728); summands (k, 9, n)."""
        k, _, n = self._shape(summands)
        _dense(n, summands)
        out, ov, err = self._cols(9, n), self._vec(n), self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_add_many_witness_batch(self._h, C.c_int(field), _ptr(summands), C.c_int(k),
                                                             _ptr(out), _ptr(ov), C.c_size_t(n), C.c_size_t(n), _ptr(err)))
        return out, ov, err, bad

    def inv_witness_batch(self, field, x):
        """NonNativeInverseGenerator (gadgets/nonnative.rs:857-872)."""
        n = self._shape(x)[1]
        _dense(n, x)
        inv, div, err = self._cols(9, n), self._cols(9, n), self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_inv_witness_batch(self._h, C.c_int(field), _ptr(x), _ptr(inv), _ptr(div),
                                                        C.c_size_t(n), C.c_size_t(n), _ptr(err)))
        return inv, div, err, bad

    def biguint_div_rem_batch(self, a, b):
        """BigUintDivRemGenerator (gadgets/biguint.rs:508-518): a (na, n), b (nb, n) limb columns ->
        (div (max(0, na - nb + 1), n), rem (nb, n), err, flagged)."""
        na, n = self._shape(a)
        nb = self._shape(b)[0]
        _dense(n, a, b)
        nd = 0 if nb > na + 1 else na - nb + 1
        div, rem, err = self._cols(max(nd, 1), n), self._cols(nb, n), self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_biguint_div_rem_batch(self._h, _ptr(a), C.c_int(na), _ptr(b), C.c_int(nb), _ptr(div),
                                                            _ptr(rem), C.c_size_t(n), C.c_size_t(n), _ptr(err)))
        return div[:nd], rem, err, bad

    def glv_decompose_batch(self, k):
        """GLVDecompositionGenerator (gadgets/glv.rs:128-142)."""
        n = self._shape(k)[1]
        _dense(n, k)
        k1, k2, n1, n2, err = self._cols(5, n), self._cols(5, n), self._vec(n), self._vec(n), self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_glv_decompose_batch(self._h, _ptr(k), _ptr(k1), _ptr(k2), _ptr(n1), _ptr(n2),
                                                          C.c_size_t(n), C.c_size_t(n), _ptr(err)))
        return k1, k2, n1, n2, err, bad

    def limb_split(self, packed, out=None):
        """set_biguint_target (gadgets/biguint.rs:454-463): (n, 32) uint8 -> (9, n) limb columns."""
        n = self._shape(packed)[0]
        limbs = out if out is not None else self._cols(9, n)
        self._check(self._L.p2e_limb_split(self._h, _ptr(packed), _ptr(limbs), C.c_size_t(n), C.c_size_t(_ld(limbs))))
        return limbs

    def limb_pack(self, limbs):
        """get_biguint_target (gadgets/biguint.rs:444-452): (9, n) limb columns -> (n, 32) uint8."""
        n = self._shape(limbs)[1]
        ld = _ld(limbs)
        if self.host_pointers:
            packed = np.zeros((n, 32), dtype=np.uint8)
        else:
            import torch
            packed = torch.empty((n, 32), dtype=torch.uint8, device=f"cuda:{self.device}")
        err = self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_limb_pack(self._h, _ptr(limbs), _ptr(packed), C.c_size_t(n), C.c_size_t(ld), _ptr(err)))
        return packed, err, bad

    def columns_to_rows(self, cols, n=None, ld=None, rows=None):
        """(ncols, ld) column-major matrix -> (n, ncols) matrix with one contiguous witness per signature."""
        ncols = self._shape(cols)[0]
        ld = ld if ld is not None else _ld(cols)
        n = n if n is not None else self._shape(cols)[1]
        if rows is None:
            if self.host_pointers:
                rows = np.zeros((n, ncols), dtype=np.uint64)
            else:
                import torch
                rows = torch.empty((n, ncols), dtype=torch.int64, device=f"cuda:{self.device}")
        self._check(self._L.p2e_columns_to_rows(self._h, _ptr(cols), C.c_size_t(ld), C.c_size_t(n), C.c_size_t(ncols),
                                                _ptr(rows), C.c_size_t(_ld(rows))))
        return rows

    def compact_to_rows(self, program, narrow, wide, n, ld_narrow=None, ld_wide=None, rows_narrow=None, rows_wide=None):
        """columns_to_rows for the compact container: ((n, num_narrow) u32, (n, num_wide) u64), one contiguous witness per signature."""
        _m, nn, nw = compact_layout(program)
        if rows_narrow is None or rows_wide is None:
            if self.host_pointers:
                rows_narrow, rows_wide = np.zeros((n, nn), dtype=np.uint32), np.zeros((n, nw), dtype=np.uint64)
            else:
                import torch
                dev = f"cuda:{self.device}"
                rows_narrow = torch.empty((n, nn), dtype=torch.int32, device=dev)
                rows_wide = torch.empty((n, nw), dtype=torch.int64, device=dev)
        self._check(self._L.p2e_compact_to_rows(self._h, C.c_int(program), _ptr(narrow), C.c_size_t(ld_narrow or _ld(narrow)),
                                                _ptr(wide), C.c_size_t(ld_wide or _ld(wide)), C.c_size_t(n),
                                                _ptr(rows_narrow), C.c_size_t(_ld(rows_narrow)), _ptr(rows_wide),
                                                C.c_size_t(_ld(rows_wide))))
        return rows_narrow, rows_wide

    # ---- fused schedules -----------------------------------------------------------------------------
    def ecdsa_verify_witness_batch(self, msg, r, s, pkx, pky, cols=None, err=None, valid=None, ld=None):
        """verify_secp256k1_message_circuit (gadgets/ecdsa.rs:30-53): (82615, n) Goldilocks columns.
        ``cols`` may be a column slice of a wider matrix: pass its row stride as ``ld``."""
        n = self._shape(msg)[0]
        if cols is None:
            # a power-of-two column stride makes consecutive columns camp on the same HBM channels (-9 %)
            ld = n + 16 if (n >= 4096 and n & (n - 1) == 0) else n
            full = self._cols(VERIFY_COLS, ld)
            cols = full[:, :n] if ld != n else full
        err = err if err is not None else self._vec(n, np.uint8)
        valid = valid if valid is not None else self._vec(n, np.uint8)
        ld = ld if ld is not None else _ld(cols)
        bad = self._check(self._L.p2e_ecdsa_verify_witness_batch(self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky),
                                                                 _ptr(cols), C.c_size_t(n), C.c_size_t(ld), _ptr(err), _ptr(valid)))
        return cols, err, valid, bad

    def aux_witness_batch(self, program, pky, cols, n=None, ld=None, aux=None, err=None, ld_aux=None):
        """Built-in-generator columns (bool selects, window bits / digits, random-access selections, is_equal / not
        results: SURVEY.md 8(f) rank 1) derived from a finished witness matrix: (8959 | 4738, n) columns."""
        n = n if n is not None else self._shape(pky)[0]
        ld = ld if ld is not None else _ld(cols)
        if aux is None:
            aux = self._cols(VERIFY_AUX_COLS if program == PROGRAM_VERIFY else GLV_MUL_AUX_COLS, n)
        ld_aux = ld_aux if ld_aux is not None else _ld(aux)
        err = err if err is not None else self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_aux_witness_batch(self._h, C.c_int(program), _ptr(pky), _ptr(cols), C.c_size_t(ld),
                                                        _ptr(aux), C.c_size_t(ld_aux), C.c_size_t(n), _ptr(err)))
        return aux, err, bad

    def ux_witness_batch(self, program, inputs, cols, aux, n=None, ld=None, ld_aux=None, ux=None, ld_ux=None, err=None, u32=True):
        """Constraint-block columns (what the U29 gates inside add / sub / add_many / inv / the
        range checks hand back, in builder-call order) from the finished witness and aux matrices.
        inputs: (msg, r, s, pk.x, pk.y) for the verify program, (pk.x, pk.y, k) for glv_mul.  (249385 | 194361, n)."""
        if program == PROGRAM_VERIFY:
            msg, r, s, pkx, pky = inputs
        else:
            pkx, pky, msg = inputs
            r = s = None
        n = n if n is not None else self._shape(pky)[0]
        ld = ld if ld is not None else _ld(cols)
        ld_aux = ld_aux if ld_aux is not None else _ld(aux)
        k = VERIFY_UX_COLS if program == PROGRAM_VERIFY else GLV_MUL_UX_COLS
        if ux is None:
            if self.host_pointers:
                ux = np.zeros((k, n), dtype=np.uint32 if u32 else np.uint64)
            else:
                import torch
                ux = torch.empty((k, n), dtype=torch.int32 if u32 else torch.int64, device=f"cuda:{self.device}")
        else:
            u32 = (ux.dtype == np.uint32) if isinstance(ux, np.ndarray) else (ux.element_size() == 4)
        ld_ux = ld_ux if ld_ux is not None else _ld(ux)
        err = err if err is not None else self._vec(n, np.uint8)
        self._L.p2e_ux_witness_batch.restype = C.c_long
        bad = self._check(self._L.p2e_ux_witness_batch(self._h, C.c_int(program), _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky),
                                                       _ptr(cols), C.c_size_t(ld), _ptr(aux), C.c_size_t(ld_aux), _ptr(ux),
                                                       C.c_int(1 if u32 else 0), C.c_size_t(ld_ux), C.c_size_t(n), _ptr(err)))
        return ux, err, bad

    # ---- wire-matrix assembly --------------------------------------------------
    def wire_map(self, program, src, dst, num_wires, degree):
        """Upload a column -> (wire, row) map (include/p2e.h p2e_wire_map_create); returns an opaque handle for
        assemble_wires.  src / dst: uint32 arrays (plonky2_ecdsa_amd.wiremap has a synthetic generator)."""
        src, dst = np.ascontiguousarray(src, np.uint32), np.ascontiguousarray(dst, np.uint32)
        ent = np.empty((len(src), 2), dtype=np.uint32)
        ent[:, 0], ent[:, 1] = src, dst
        h = C.c_void_p()
        rc = self._L.p2e_wire_map_create(self._h, C.c_int(program), _ptr(ent), C.c_size_t(len(src)), C.c_uint32(num_wires),
                                         C.c_uint32(degree), C.byref(h))
        if rc != 0:
            raise P2EError(f"p2e_wire_map_create failed ({rc}): {self._L.p2e_last_error().decode()}")
        return _WireMap(self, h, program, num_wires, degree, len(src))

    def gate_internal_batch(self, program, aux, n=None, ld_aux=None, gate=None, ld_gate=None):
        """Gate-internal values of the built-in gates (equality gadget internals, RandomAccessGate index bits) from the
        aux matrix: (10703 | 5621, n) uint64."""
        n = n if n is not None else self._shape(aux)[1]
        ld_aux = ld_aux if ld_aux is not None else _ld(aux)
        if gate is None:
            gate = self._cols(VERIFY_GATE_COLS if program == PROGRAM_VERIFY else GLV_MUL_GATE_COLS, n)
        ld_gate = ld_gate if ld_gate is not None else _ld(gate)
        self._L.p2e_gate_internal_batch.restype = C.c_long
        self._check(self._L.p2e_gate_internal_batch(self._h, C.c_int(program), _ptr(aux), C.c_size_t(ld_aux), _ptr(gate),
                                                    C.c_size_t(ld_gate), C.c_size_t(n)))
        return gate

    def assemble_wires(self, wmap, cols=None, aux=None, ux=None, gate=None, wires=None, n=None):
        """wires[i, wire * degree + row] = the mapped value of signature i: (n, num_wires * degree) int64, one plonky2 wire
        matrix per signature (zero where the map names nothing, when allocated here)."""
        ref = next(m for m in (cols, aux, ux, gate) if m is not None)
        n = n if n is not None else self._shape(ref)[1]
        cells = wmap.num_wires * wmap.degree
        if wires is None:
            if self.host_pointers:
                wires = np.zeros((n, cells), dtype=np.uint64)
            else:
                import torch
                wires = torch.zeros((n, cells), dtype=torch.int64, device=f"cuda:{self.device}")
        u32 = 0
        if ux is not None:
            u32 = int((ux.dtype == np.uint32) if isinstance(ux, np.ndarray) else (ux.element_size() == 4))
        self._L.p2e_assemble_wires.restype = C.c_long
        self._check(self._L.p2e_assemble_wires(self._h, wmap._h, _ptr(cols), C.c_size_t(_ld(cols) if cols is not None else 0),
                                               _ptr(aux), C.c_size_t(_ld(aux) if aux is not None else 0), _ptr(ux), C.c_int(u32),
                                               C.c_size_t(_ld(ux) if ux is not None else 0), _ptr(gate),
                                               C.c_size_t(_ld(gate) if gate is not None else 0), _ptr(wires),
                                               C.c_size_t(_ld(wires)), C.c_size_t(n)))
        return wires

    def columns_compact(self, program, cols, n=None, ld=None, narrow=None, wide=None, err=None, ld_narrow=None, ld_wide=None):
        """Repack a finished witness matrix for transfers: (narrow u32 (num_narrow, n), wide u64 (num_wide, n), err, bad).
        Preallocated `narrow` / `wide` may be column slices of wider matrices: pass their row strides."""
        _m, nn, nw = compact_layout(program)
        ld = ld if ld is not None else _ld(cols)
        n = n if n is not None else self._shape(cols)[1]
        if narrow is None or wide is None:
            if self.host_pointers:
                narrow, wide = np.zeros((nn, n), dtype=np.uint32), np.zeros((nw, n), dtype=np.uint64)
            else:
                import torch
                dev = f"cuda:{self.device}"
                narrow = torch.empty((nn, n), dtype=torch.int32, device=dev)
                wide = torch.empty((nw, n), dtype=torch.int64, device=dev)
        err = err if err is not None else self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_columns_compact(self._h, C.c_int(program), _ptr(cols), C.c_size_t(ld), C.c_size_t(n),
                                                      _ptr(narrow), C.c_size_t(ld_narrow or _ld(narrow)), _ptr(wide),
                                                      C.c_size_t(ld_wide or _ld(wide)), _ptr(err)))
        return narrow, wide, err, bad

    def ecdsa_verify_batch(self, msg, r, s, pkx, pky, err=None, valid=None):
        """The verdict alone (pre-filter, no witness): (err, valid, flagged count), as ecdsa_verify_witness_batch sets them."""
        n = self._shape(msg)[0]
        err = err if err is not None else self._vec(n, np.uint8)
        valid = valid if valid is not None else self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_ecdsa_verify_batch(self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky),
                                                         C.c_size_t(n), _ptr(err), _ptr(valid)))
        return err, valid, bad

    def _compact_out(self, program, n, narrow, wide, ld_narrow, ld_wide):
        _m, nn, nw = compact_layout(program)
        if narrow is None or wide is None:
            ldp = n + 16 if (not self.host_pointers and n >= 4096 and n & (n - 1) == 0) else n   # see ecdsa_verify_witness_batch
            if self.host_pointers:
                narrow, wide = np.zeros((nn, ldp), dtype=np.uint32), np.zeros((nw, ldp), dtype=np.uint64)
            else:
                import torch
                dev = f"cuda:{self.device}"
                narrow = torch.empty((nn, ldp), dtype=torch.int32, device=dev)
                wide = torch.empty((nw, ldp), dtype=torch.int64, device=dev)
            # same convention as ecdsa_verify_witness_batch: (k, n) views whose row stride is the padded one
            narrow, wide = narrow[:, :n], wide[:, :n]
        return narrow, wide, ld_narrow or _ld(narrow), ld_wide or _ld(wide)

    def ecdsa_verify_witness_compact_batch(self, msg, r, s, pkx, pky, narrow=None, wide=None, err=None, valid=None,
                                           ld_narrow=None, ld_wide=None):
        """ecdsa_verify_witness_batch writing the compact container (include/p2e.h): (narrow u32, wide u64, err, valid, bad)."""
        n = self._shape(msg)[0]
        narrow, wide, ldn, ldw = self._compact_out(PROGRAM_VERIFY, n, narrow, wide, ld_narrow, ld_wide)
        err = err if err is not None else self._vec(n, np.uint8)
        valid = valid if valid is not None else self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_ecdsa_verify_witness_compact_batch(
            self._h, _ptr(msg), _ptr(r), _ptr(s), _ptr(pkx), _ptr(pky), _ptr(narrow), C.c_size_t(ldn), _ptr(wide), C.c_size_t(ldw),
            C.c_size_t(n), _ptr(err), _ptr(valid)))
        return narrow, wide, err, valid, bad

    def glv_mul_witness_compact_batch(self, px, py, k, narrow=None, wide=None, err=None, valid=None, ld_narrow=None, ld_wide=None):
        n = self._shape(px)[0]
        narrow, wide, ldn, ldw = self._compact_out(PROGRAM_GLV_MUL, n, narrow, wide, ld_narrow, ld_wide)
        err = err if err is not None else self._vec(n, np.uint8)
        valid = valid if valid is not None else self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_glv_mul_witness_compact_batch(
            self._h, _ptr(px), _ptr(py), _ptr(k), _ptr(narrow), C.c_size_t(ldn), _ptr(wide), C.c_size_t(ldw), C.c_size_t(n),
            _ptr(err), _ptr(valid)))
        return narrow, wide, err, valid, bad

    def aux_witness_compact_batch(self, program, pky, narrow, n=None, ld_narrow=None, aux32=None, err=None, ld_aux=None):
        """aux_witness_batch inside the compact container: reads the narrow u32 matrix, writes a u32 aux matrix."""
        n = n if n is not None else self._shape(pky)[0]
        ld_narrow = ld_narrow if ld_narrow is not None else _ld(narrow)
        if aux32 is None:
            k = VERIFY_AUX_COLS if program == PROGRAM_VERIFY else GLV_MUL_AUX_COLS
            if self.host_pointers:
                aux32 = np.zeros((k, n), dtype=np.uint32)
            else:
                import torch
                aux32 = torch.empty((k, n), dtype=torch.int32, device=f"cuda:{self.device}")
        ld_aux = ld_aux if ld_aux is not None else _ld(aux32)
        err = err if err is not None else self._vec(n, np.uint8)
        bad = self._check(self._L.p2e_aux_witness_compact_batch(self._h, C.c_int(program), _ptr(pky), _ptr(narrow),
                                                                C.c_size_t(ld_narrow), _ptr(aux32), C.c_size_t(ld_aux),
                                                                C.c_size_t(n), _ptr(err)))
        return aux32, err, bad

    def glv_mul_witness_batch(self, px, py, k, cols=None, err=None, valid=None, ld=None):
        """glv_mul (gadgets/glv.rs:87-104): (65243, n) Goldilocks columns."""
        n = self._shape(px)[0]
        cols = cols if cols is not None else self._cols(GLV_MUL_COLS, n)
        err = err if err is not None else self._vec(n, np.uint8)
        valid = valid if valid is not None else self._vec(n, np.uint8)
        ld = ld if ld is not None else _ld(cols)
        bad = self._check(self._L.p2e_glv_mul_witness_batch(self._h, _ptr(px), _ptr(py), _ptr(k), _ptr(cols),
                                                            C.c_size_t(n), C.c_size_t(ld), _ptr(err), _ptr(valid)))
        return cols, err, valid, bad
