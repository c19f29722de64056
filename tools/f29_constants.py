"""The constants of csrc/fe29.hpp, derived and checked (no GPU needed):
  * 2^261 mod p = R0 + 2^R1S * 2^29 for secp256k1's p,
  * the multiples K p written with "borrowed" limbs so that every limb lies in [m T, m T + 2^29) for the subtrahend
    classes m = 1 .. 4 (T = 2^29 + 2^20): a - b is computed limb by limb as a + (K p - b),
and compared with the values in the header."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**256 - 2**32 - 977
M = (1 << 29) - 1
T = (1 << 29) + (1 << 20)
r = pow(2, 261, P)
R0, R1 = r & M, r >> 29
assert r == R0 + (R1 << 29) and R1 == 1 << 8, (R0, R1)
print(f"2^261 mod p = {R0} + 2^8 * 2^29   (F29_R0 = {R0}, F29_R1S = 8)")


def borrowed(K, lo):
    v, out = K * P, []
    for _ in range(8):
        l = v & M
        v >>= 29
        while l < lo:
            l += 1 << 29
            v -= 1
        out.append(l)
    out.append(v)
    return out


rows = []
for m in (1, 2, 3, 4):
    K = next(K for K in range(1, 4096) if borrowed(K, m * T)[8] >= m * T)
    L = borrowed(K, m * T)
    assert sum(l << (29 * k) for k, l in enumerate(L)) == K * P and all(m * T <= l < m * T + (1 << 29) for l in L) and max(L) < 2**32
    assert L[2:8] == [L[2]] * 6
    rows.append(L)
    print(f"class {m}: K = {K:3d}  limbs 0, 1, 2..7, 8 = {L[0]:#x} {L[1]:#x} {L[2]:#x} {L[8]:#x}")
src = open(os.path.join(ROOT, "plonky2-ecdsa_amd", "csrc", "fe29.hpp")).read()
for name, idx in (("c0", 0), ("c1", 1), ("cm", 2), ("c8", 8)):
    vals = [int(x, 16) for x in re.search(r"const u32 %s\[4\] = \{([^}]*)\}" % name, src).group(1).replace("u", "").split(",")]
    assert vals == [L[idx] for L in rows], (name, vals)
assert f"F29_R0 = {R0}u" in src
print("csrc/fe29.hpp agrees")
