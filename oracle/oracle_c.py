"""ctypes binding of oracle/libp2e_oracle.so (TEST INFRASTRUCTURE ONLY -- see p2e_oracle.h).

numpy in / numpy out; column-major Goldilocks columns ``[col][n]`` (C-contiguous ``(ncols, n)`` uint64
arrays), packed 256-bit values as ``(n, 32)`` uint8 little-endian.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VERIFY_COLS = 82615
GLV_MUL_COLS = 65243
VERIFY_AUX = 8959
GLV_MUL_AUX = 4738


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libp2e_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libp2e_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        for name in ("mul_witness", "checksum_witness", "add_witness", "sub_witness", "add_many_witness",
                     "inv_witness", "glv_decompose", "limb_split", "limb_pack", "verify_witness",
                     "glv_mul_witness", "verify_witness_aux", "glv_mul_witness_aux", "div_rem"):
            getattr(_LIB, "p2e_oracle_" + name).restype = C.c_long
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _cols(k, n):
    return np.zeros((k, n), dtype=np.uint64)


def _chk(a, k):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[0] == k, a.shape
    return a


def mul_witness(field, x, y):
    x, y = _chk(x, 9), _chk(y, 9)
    n = x.shape[1]
    r, q, cs, b = _cols(9, n), _cols(9, n), _cols(17, n), _cols(16, n)
    err = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_mul_witness(C.c_int(field), _p(x), _p(y), _p(r), _p(q), _p(cs), _p(b), C.c_size_t(n),
                                 C.c_size_t(n), _p(err))
    return r, q, cs, b, err


def checksum_witness(a):
    a = _chk(a, 17)
    n = a.shape[1]
    b = _cols(16, n)
    err = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_checksum_witness(_p(a), _p(b), C.c_size_t(n), C.c_size_t(n), _p(err))
    return b, err


def _binop(fn, field, a, b):
    a, b = _chk(a, 9), _chk(b, 9)
    n = a.shape[1]
    out, ov = _cols(9, n), np.zeros(n, dtype=np.uint64)
    err = np.zeros(n, dtype=np.uint8)
    fn(C.c_int(field), _p(a), _p(b), _p(out), _p(ov), C.c_size_t(n), C.c_size_t(n), _p(err))
    return out, ov, err


def add_witness(field, a, b):
    return _binop(lib().p2e_oracle_add_witness, field, a, b)


def sub_witness(field, a, b):
    return _binop(lib().p2e_oracle_sub_witness, field, a, b)


def add_many_witness(field, summands):
    s = np.ascontiguousarray(summands, dtype=np.uint64)
    assert s.ndim == 3 and s.shape[1] == 9
    k, _, n = s.shape
    out, ov = _cols(9, n), np.zeros(n, dtype=np.uint64)
    err = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_add_many_witness(C.c_int(field), _p(s), C.c_int(k), _p(out), _p(ov), C.c_size_t(n),
                                      C.c_size_t(n), _p(err))
    return out, ov, err


def inv_witness(field, x):
    x = _chk(x, 9)
    n = x.shape[1]
    inv, div = _cols(9, n), _cols(9, n)
    err = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_inv_witness(C.c_int(field), _p(x), _p(inv), _p(div), C.c_size_t(n), C.c_size_t(n), _p(err))
    return inv, div, err


def glv_decompose(k):
    k = _chk(k, 9)
    n = k.shape[1]
    k1, k2 = _cols(5, n), _cols(5, n)
    n1, n2 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    err = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_glv_decompose(_p(k), _p(k1), _p(k2), _p(n1), _p(n2), C.c_size_t(n), C.c_size_t(n), _p(err))
    return k1, k2, n1, n2, err


def limb_split(packed):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n = packed.shape[0]
    out = _cols(9, n)
    lib().p2e_oracle_limb_split(_p(packed), _p(out), C.c_size_t(n), C.c_size_t(n))
    return out


def limb_pack(limbs):
    limbs = _chk(limbs, 9)
    n = limbs.shape[1]
    out = np.zeros((n, 32), dtype=np.uint8)
    err = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_limb_pack(_p(limbs), _p(out), C.c_size_t(n), C.c_size_t(n), _p(err))
    return out, err


def div_rem(a, b):
    """BigUintDivRemGenerator: a (na, n), b (nb, n) limb columns -> (div (max(0, na-nb+1), n), rem (nb, n), err)."""
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    na, nb, n = a.shape[0], b.shape[0], a.shape[1]
    nd = 0 if nb > na + 1 else na - nb + 1
    div, rem = _cols(max(nd, 1), n), _cols(nb, n)
    err = np.zeros(n, dtype=np.uint8)
    rc = lib().p2e_oracle_div_rem(_p(a), C.c_int(na), _p(b), C.c_int(nb), _p(div), _p(rem), C.c_size_t(n), C.c_size_t(n), _p(err))
    assert rc >= 0
    return div[:nd], rem, err


def verify_witness(msg, r, s, pkx, pky, nthreads=0):
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (msg, r, s, pkx, pky)]
    n = arrs[0].shape[0]
    cols = _cols(VERIFY_COLS, n)
    err = np.zeros(n, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_verify_witness(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(err),
                                    _p(flags), C.c_int(nthreads))
    return cols, err, flags


def verify_witness_lockstep(msg, r, s, pkx, pky, nthreads=0, group=64):
    """verify_witness by the optimised-CPU variant: lock-step groups, one Fermat ladder per group and inverse op
    (p2e_oracle.h p2e_oracle_verify_witness_lockstep).  Same columns, bit for bit."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (msg, r, s, pkx, pky)]
    n = arrs[0].shape[0]
    cols = _cols(VERIFY_COLS, n)
    err = np.zeros(n, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    f = lib().p2e_oracle_verify_witness_lockstep
    f.restype = C.c_long
    rc = f(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(err), _p(flags), C.c_int(nthreads), C.c_int(group))
    if rc < 0:
        raise MemoryError("p2e_oracle_verify_witness_lockstep: coroutine stacks")
    return cols, err, flags


def verify_witness_aux_lockstep(msg, r, s, pkx, pky, nthreads=0, group=64):
    """(cols, aux, err, flags) by the lock-step walk: hot-path columns plus the built-in-generator values."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (msg, r, s, pkx, pky)]
    n = arrs[0].shape[0]
    cols, aux = _cols(VERIFY_COLS, n), _cols(VERIFY_AUX, n)
    err = np.zeros(n, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    f = lib().p2e_oracle_verify_witness_aux_lockstep
    f.restype = C.c_long
    rc = f(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(aux), C.c_size_t(n), _p(err), _p(flags),
           C.c_int(nthreads), C.c_int(group))
    if rc < 0:
        raise MemoryError("p2e_oracle_verify_witness_aux_lockstep: coroutine stacks")
    return cols, aux, err, flags


def glv_mul_witness(px, py, k, nthreads=0):
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (px, py, k)]
    n = arrs[0].shape[0]
    cols = _cols(GLV_MUL_COLS, n)
    err = np.zeros(n, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_glv_mul_witness(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(err),
                                     _p(flags), C.c_int(nthreads))
    return cols, err, flags


def verify_witness_aux(msg, r, s, pkx, pky, nthreads=0):
    """(cols, aux, err, flags): hot-path columns plus the built-in-generator values (p2e_oracle.h)."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (msg, r, s, pkx, pky)]
    n = arrs[0].shape[0]
    cols, aux = _cols(VERIFY_COLS, n), _cols(VERIFY_AUX, n)
    err = np.zeros(n, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_verify_witness_aux(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(aux),
                                        C.c_size_t(n), _p(err), _p(flags), C.c_int(nthreads))
    return cols, aux, err, flags


def glv_mul_witness_aux(px, py, k, nthreads=0):
    arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (px, py, k)]
    n = arrs[0].shape[0]
    cols, aux = _cols(GLV_MUL_COLS, n), _cols(GLV_MUL_AUX, n)
    err = np.zeros(n, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    lib().p2e_oracle_glv_mul_witness_aux(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(aux),
                                         C.c_size_t(n), _p(err), _p(flags), C.c_int(nthreads))
    return cols, aux, err, flags


CP_WINDOWED_MUL, CP_SCALAR_MUL, CP_VERIFY = 1, 2, 3


def curve_program_num_cols(kind, curve):
    """(num_cols, num_aux) of a curve program, by a dry walk (p2e_oracle.h p2e_oracle_curve_program_num_cols)."""
    f = lib().p2e_oracle_curve_program_num_cols
    f.restype = C.c_long
    na = C.c_long()
    nc = f(C.c_int(kind), C.c_int(curve), C.byref(na))
    assert nc > 0
    return int(nc), int(na.value)


def curve_program(kind, curve, blind, args, nthreads=0, lockstep=0, want_aux=True):
    """The crate's other scalar multiplications (p2e_oracle.h p2e_oracle_curve_program).  blind: (x, y) python ints or
    two 32-byte arrays; args: (px, py, k) for kinds 1, 2 or (msg, r, s, pkx, pky) for kind 3, each (n, 32) uint8.
    Returns (cols, aux, err, flags)."""
    bx, by = [np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8).copy() if isinstance(v, int) else
              np.ascontiguousarray(v, dtype=np.uint8) for v in blind]
    if len(args) == 3:
        px, py, k = [np.ascontiguousarray(a, dtype=np.uint8) for a in args]
        msg = r = s = k
    else:
        msg, r, s, px, py = [np.ascontiguousarray(a, dtype=np.uint8) for a in args]
    n = px.shape[0]
    nc, na = curve_program_num_cols(kind, curve)
    cols = _cols(nc, n)
    aux = _cols(na, n) if want_aux else None
    err, flags = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    f = lib().p2e_oracle_curve_program
    f.restype = C.c_long
    rc = f(C.c_int(kind), C.c_int(curve), _p(bx), _p(by), _p(msg), _p(r), _p(s), _p(px), _p(py), _p(cols), C.c_size_t(n),
           C.c_size_t(n), _p(aux) if want_aux else None, C.c_size_t(n), _p(err), _p(flags), C.c_int(nthreads), C.c_int(lockstep))
    if rc < 0:
        raise RuntimeError(f"p2e_oracle_curve_program: {rc}")
    return cols, aux, err, flags


def rando():
    x = np.zeros(32, dtype=np.uint8)
    y = np.zeros(32, dtype=np.uint8)
    lib().p2e_oracle_rando(_p(x), _p(y))
    return int.from_bytes(x.tobytes(), "little"), int.from_bytes(y.tobytes(), "little")


def max_threads():
    return int(lib().p2e_oracle_max_threads())


# helpers shared by tests
def pack256(vals):
    """list of python ints -> (n, 32) uint8 little-endian"""
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def unpack256(arr):
    return [int.from_bytes(bytes(row), "little") for row in np.asarray(arr, dtype=np.uint8)]


def limbs_cols(vals, nl=9):
    """list of python ints -> (nl, n) uint64 29-bit limb columns"""
    out = np.zeros((nl, len(vals)), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(nl):
            out[k, i] = (int(v) >> (29 * k)) & ((1 << 29) - 1)
    return out
