"""Three implementations behind one numpy interface, so the same parity checks run on each:
  OracleBackend  oracle/libp2e_oracle.so      (the checker)
  EmuBackend     tests/emu/libp2e_emu.so      (the HIP kernels' per-lane bodies compiled for the CPU)
  GpuBackend     plonky2-ecdsa_amd/libp2e_hip.so through the C ABI (the product; needs a GPU)
All arrays: Goldilocks columns (k, n) uint64; packed values (n, 32) uint8."""
import ctypes as C
import os

import numpy as np

import oracle_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    name = "oracle"
    mul = staticmethod(lambda f, x, y: oracle_c.mul_witness(f, x, y))
    checksum = staticmethod(oracle_c.checksum_witness)
    add = staticmethod(oracle_c.add_witness)
    sub = staticmethod(oracle_c.sub_witness)
    add_many = staticmethod(oracle_c.add_many_witness)
    inv = staticmethod(oracle_c.inv_witness)
    glv = staticmethod(oracle_c.glv_decompose)
    split = staticmethod(oracle_c.limb_split)
    pack = staticmethod(oracle_c.limb_pack)
    verify = staticmethod(lambda *a: oracle_c.verify_witness(*a))
    glv_mul = staticmethod(lambda *a: oracle_c.glv_mul_witness(*a))
    div_rem = staticmethod(lambda a, b: oracle_c.div_rem(a, b))

    @staticmethod
    def aux(program, inputs):
        """(cols, aux, err): the oracle's own walk records both matrices."""
        f = oracle_c.verify_witness_aux if program == 0 else oracle_c.glv_mul_witness_aux
        cols, aux, err, _flags = f(*inputs)
        return cols, aux, err


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _z(k, n):
    return np.zeros((k, n), dtype=np.uint64)


class EmuBackend:
    name = "emu"

    def __init__(self):
        self.L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libp2e_emu.so"))
        for f in ("emu_verify", "emu_glv_mul", "emu_mul", "emu_checksum", "emu_add", "emu_sub", "emu_add_many",
                  "emu_inv", "emu_glv", "emu_split", "emu_pack", "emu_aux", "emu_aux_num_cols", "emu_verify_compact",
                  "emu_glv_mul_compact", "emu_aux_compact", "emu_verify_only", "emu_div_rem", "emu_ux", "emu_gate"):
            getattr(self.L, f).restype = C.c_long

    def gate(self, program, aux):
        """gate-internal values of the built-in gates from an aux matrix: (num_gate_cols, n) uint64"""
        aux = np.ascontiguousarray(aux, np.uint64)
        n = aux.shape[1]
        k = 10703 if program == 0 else 5621
        out = _z(k, n)
        assert self.L.emu_gate(C.c_int(program), _p(aux), C.c_size_t(n), _p(out), C.c_size_t(n), C.c_size_t(n)) == k
        return out

    def ux(self, program, inputs, cols, aux):
        """constraint-block columns from finished matrices: (num_ux_cols, n) uint64, err"""
        if program == 0:
            msg, r, s, pkx, pky = [np.ascontiguousarray(a, np.uint8) for a in inputs]
        else:
            pkx, pky, msg = [np.ascontiguousarray(a, np.uint8) for a in inputs]
            r = s = None
        cols, aux = np.ascontiguousarray(cols, np.uint64), np.ascontiguousarray(aux, np.uint64)
        n = cols.shape[1]
        k = 249385 if program == 0 else 194361
        ux, err = _z(k, n), np.zeros(n, np.uint8)
        got = self.L.emu_ux(C.c_int(program), _p(msg), _p(r) if r is not None else None, _p(s) if s is not None else None, _p(pkx),
                            _p(pky), _p(cols), C.c_size_t(n), _p(aux), C.c_size_t(n), _p(ux), C.c_size_t(n), C.c_size_t(n), _p(err))
        assert got == k
        return ux, err

    def mul(self, field, x, y):
        x, y = np.ascontiguousarray(x, np.uint64), np.ascontiguousarray(y, np.uint64)
        n = x.shape[1]
        r, q, cs, b, err = _z(9, n), _z(9, n), _z(17, n), _z(16, n), np.zeros(n, np.uint8)
        self.L.emu_mul(C.c_int(field), _p(x), _p(y), _p(r), _p(q), _p(cs), _p(b), C.c_size_t(n), C.c_size_t(n), _p(err))
        return r, q, cs, b, err

    def div_rem(self, a, b):
        a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
        na, nb, n = a.shape[0], b.shape[0], a.shape[1]
        nd = 0 if nb > na + 1 else na - nb + 1
        div, rem, err = _z(max(nd, 1), n), _z(nb, n), np.zeros(n, np.uint8)
        self.L.emu_div_rem(_p(a), C.c_int(na), _p(b), C.c_int(nb), _p(div), _p(rem), C.c_size_t(n), C.c_size_t(n), _p(err))
        return div[:nd], rem, err

    def checksum(self, a):
        a = np.ascontiguousarray(a, np.uint64)
        n = a.shape[1]
        b, err = _z(16, n), np.zeros(n, np.uint8)
        self.L.emu_checksum(_p(a), _p(b), C.c_size_t(n), C.c_size_t(n), _p(err))
        return b, err

    def _bin(self, fn, field, a, b):
        a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
        n = a.shape[1]
        out, ov, err = _z(9, n), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        fn(C.c_int(field), _p(a), _p(b), _p(out), _p(ov), C.c_size_t(n), C.c_size_t(n), _p(err))
        return out, ov, err

    def add(self, field, a, b):
        return self._bin(self.L.emu_add, field, a, b)

    def sub(self, field, a, b):
        return self._bin(self.L.emu_sub, field, a, b)

    def add_many(self, field, s):
        s = np.ascontiguousarray(s, np.uint64)
        k, _, n = s.shape
        out, ov, err = _z(9, n), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        self.L.emu_add_many(C.c_int(field), _p(s), C.c_int(k), _p(out), _p(ov), C.c_size_t(n), C.c_size_t(n), _p(err))
        return out, ov, err

    def inv(self, field, x):
        x = np.ascontiguousarray(x, np.uint64)
        n = x.shape[1]
        inv, div, err = _z(9, n), _z(9, n), np.zeros(n, np.uint8)
        self.L.emu_inv(C.c_int(field), _p(x), _p(inv), _p(div), C.c_size_t(n), C.c_size_t(n), _p(err))
        return inv, div, err

    def glv(self, k):
        k = np.ascontiguousarray(k, np.uint64)
        n = k.shape[1]
        k1, k2, n1, n2, err = _z(5, n), _z(5, n), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        self.L.emu_glv(_p(k), _p(k1), _p(k2), _p(n1), _p(n2), C.c_size_t(n), C.c_size_t(n), _p(err))
        return k1, k2, n1, n2, err

    def split(self, packed):
        packed = np.ascontiguousarray(packed, np.uint8)
        n = packed.shape[0]
        out = _z(9, n)
        self.L.emu_split(_p(packed), _p(out), C.c_size_t(n), C.c_size_t(n))
        return out

    def pack(self, limbs):
        limbs = np.ascontiguousarray(limbs, np.uint64)
        n = limbs.shape[1]
        out, err = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
        self.L.emu_pack(_p(limbs), _p(out), C.c_size_t(n), C.c_size_t(n), _p(err))
        return out, err

    def verify(self, msg, r, s, pkx, pky, chunk=512, run_iters=4):
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (msg, r, s, pkx, pky)]
        n = arrs[0].shape[0]
        cols, err, valid = _z(82615, n), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        self.L.emu_verify(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(err), _p(valid), C.c_int(chunk),
                          C.c_int(run_iters))
        return cols, err, valid

    def glv_mul(self, px, py, k, chunk=512, run_iters=4):
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (px, py, k)]
        n = arrs[0].shape[0]
        cols, err, valid = _z(65243, n), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        self.L.emu_glv_mul(*[_p(a) for a in arrs], _p(cols), C.c_size_t(n), C.c_size_t(n), _p(err), _p(valid), C.c_int(chunk),
                           C.c_int(run_iters))
        return cols, err, valid


def _emu_aux(self, program, inputs):
    """hot-path columns from this backend's own pipeline, then the aux pass over them"""
    cols, err, _valid = (self.verify if program == 0 else self.glv_mul)(*inputs)
    cols = np.ascontiguousarray(np.asarray(cols).view(np.uint64))
    n = cols.shape[1]
    pky = np.ascontiguousarray(inputs[4] if program == 0 else inputs[1], np.uint8)
    aux = _z(int(self.L.emu_aux_num_cols(C.c_int(program))), n)
    aerr = np.zeros(n, np.uint8)
    self.L.emu_aux(C.c_int(program), _p(pky), _p(cols), C.c_size_t(n), _p(aux), C.c_size_t(n), C.c_size_t(n), _p(aerr))
    return cols, aux, np.asarray(err) | aerr


EmuBackend.aux = _emu_aux


def _emu_compact(self, program, inputs, nn, nw, run_iters=4):
    """the fused walk writing the compact container directly (CompactEmit): (narrow u32, wide u64, err, valid)"""
    arrs = [np.ascontiguousarray(a, np.uint8) for a in inputs]
    n = arrs[0].shape[0]
    narrow, wide = np.zeros((nn, n), np.uint32), np.zeros((nw, n), np.uint64)
    err, valid = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    f = self.L.emu_verify_compact if program == 0 else self.L.emu_glv_mul_compact
    f(*[_p(a) for a in arrs], _p(narrow), C.c_size_t(n), _p(wide), C.c_size_t(n), C.c_size_t(n), _p(err), _p(valid),
      C.c_int(run_iters))
    return narrow, wide, err, valid


EmuBackend.compact = _emu_compact


def _emu_aux_compact(self, program, pky, narrow):
    n = narrow.shape[1]
    aux = np.zeros((int(self.L.emu_aux_num_cols(C.c_int(program))), n), np.uint32)
    err = np.zeros(n, np.uint8)
    pky = np.ascontiguousarray(pky, np.uint8)
    self.L.emu_aux_compact(C.c_int(program), _p(pky), _p(narrow), C.c_size_t(n), _p(aux), C.c_size_t(n), C.c_size_t(n), _p(err))
    return aux, err


EmuBackend.aux_compact = _emu_aux_compact


def _emu_verify_only(self, msg, r, s, pkx, pky):
    arrs = [np.ascontiguousarray(a, np.uint8) for a in (msg, r, s, pkx, pky)]
    n = arrs[0].shape[0]
    err, valid = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    self.L.emu_verify_only(*[_p(a) for a in arrs], C.c_size_t(n), _p(err), _p(valid))
    return err, valid


EmuBackend.verify_only = _emu_verify_only


class GpuBackend:
    """The product, through the C ABI, host-pointer mode (numpy in / numpy out)."""
    name = "gpu"

    def __init__(self):
        import plonky2_ecdsa_amd as p2e
        self.ctx = p2e.Context(device=0, host_pointers=True)

    @staticmethod
    def _c(a, dt=np.uint64):
        return np.ascontiguousarray(a, dt)

    def mul(self, field, x, y):
        return self.ctx.mul_witness_batch(field, self._c(x), self._c(y))[:5]

    def checksum(self, a):
        return self.ctx.checksum_witness_batch(self._c(a))[:2]

    def add(self, field, a, b):
        return self.ctx.add_witness_batch(field, self._c(a), self._c(b))[:3]

    def sub(self, field, a, b):
        return self.ctx.sub_witness_batch(field, self._c(a), self._c(b))[:3]

    def add_many(self, field, s):
        return self.ctx.add_many_witness_batch(field, self._c(s))[:3]

    def inv(self, field, x):
        return self.ctx.inv_witness_batch(field, self._c(x))[:3]

    def glv(self, k):
        return self.ctx.glv_decompose_batch(self._c(k))[:5]

    def div_rem(self, a, b):
        return self.ctx.biguint_div_rem_batch(self._c(a), self._c(b))[:3]

    def split(self, packed):
        return self.ctx.limb_split(self._c(packed, np.uint8))

    def pack(self, limbs):
        return self.ctx.limb_pack(self._c(limbs))[:2]

    def verify(self, msg, r, s, pkx, pky):
        return self.ctx.ecdsa_verify_witness_batch(*[self._c(a, np.uint8) for a in (msg, r, s, pkx, pky)])[:3]

    def glv_mul(self, px, py, k):
        return self.ctx.glv_mul_witness_batch(*[self._c(a, np.uint8) for a in (px, py, k)])[:3]

    def aux(self, program, inputs):
        cols, err, _valid = (self.verify if program == 0 else self.glv_mul)(*inputs)
        pky = self._c(inputs[4] if program == 0 else inputs[1], np.uint8)
        aux, aerr, _bad = self.ctx.aux_witness_batch(program, pky, cols)
        return np.asarray(cols).view(np.uint64), np.asarray(aux).view(np.uint64), np.asarray(err) | np.asarray(aerr)
