"""Stand-alone generator kernels: achieved HBM GB/s (algorithmic bytes / HIP-event time).
  limb split  104 B/elem (32 in + 72 out)   -- north star: >= 40 % of HBM peak
  mul witness 552 B/elem (144 in + 408 out) -- BASELINE config 2 shape, swept up in n
Prints one JSON line per measurement."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import plonky2_ecdsa_amd as p2e

PEAK = 8000.0
ctx = p2e.Context(device=0)
reps = int(os.environ.get("REPS", "10"))


def timed(fn):
    """median wall time per call (ms).  Every call is synchronous; the median rather than the mean because the
    runtime occasionally stalls a call of a tiny batch for ~100 ms (seen once in ~10 calls at n = 2^10)."""
    import time
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


which = sys.argv[1:] or ["split", "mul", "transpose", "glv", "aux", "compact", "verdict"]
if "split" in which:
    for lg in (20, 24, 26):
        n = 1 << lg
        packed = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
        limbs = torch.empty((9, n), dtype=torch.int64, device="cuda")
        ms = timed(lambda: ctx.limb_split(packed, out=limbs))
        gbs = n * 104 / ms / 1e6
        print(json.dumps({"kernel": "k_split", "n": n, "ms": round(ms, 4), "alg_bytes": n * 104,
                          "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4)}), flush=True)
        del packed, limbs
if "mul" in which:
    for lg in (12, 16, 20, 22):
        n = 1 << lg
        x = torch.randint(0, 1 << 29, (9, n), dtype=torch.int64, device="cuda")
        y = torch.randint(0, 1 << 29, (9, n), dtype=torch.int64, device="cuda")
        x[8] &= (1 << 24) - 1
        y[8] &= (1 << 24) - 1
        for field in (0, 1):
            ms = timed(lambda: ctx.mul_witness_batch(field, x, y))
            gbs = n * 552 / ms / 1e6
            print(json.dumps({"kernel": "k_mul", "field": field, "n": n, "ms": round(ms, 4), "alg_bytes": n * 552,
                              "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 4),
                              "note": "includes output allocation by the python wrapper"}), flush=True)
        del x, y
if "transpose" in which:
    n = 1 << 14
    cols = torch.randint(0, 1 << 62, (p2e.VERIFY_COLS, n + 16), dtype=torch.int64, device="cuda")
    rows = torch.empty((n, p2e.VERIFY_COLS), dtype=torch.int64, device="cuda")
    ms = timed(lambda: ctx.columns_to_rows(cols, n=n, ld=n + 16, rows=rows))
    b = 2 * 8 * n * p2e.VERIFY_COLS
    print(json.dumps({"kernel": "k_transpose", "n": n, "ncols": p2e.VERIFY_COLS, "ms": round(ms, 4), "alg_bytes": b,
                      "GBps": round(b / ms / 1e6, 1), "frac_hbm_peak": round(b / ms / 1e6 / PEAK, 4)}), flush=True)
if "glv" in which:
    # BASELINE config 3: glv_mul witness fills (65 243 columns each), 2^10 (latency-bound) and 2^16
    import plonky2_ecdsa_amd as _p
    for lg in (10, 16):
        n = 1 << lg
        sig = _p.synth_signatures(seed=3, n=n)
        px, py, k = [torch.from_numpy(a).cuda() for a in (sig[3], sig[4], sig[0])]
        ld = n + 16
        cols = torch.empty((_p.GLV_MUL_COLS, ld), dtype=torch.int64, device="cuda")
        ms = timed(lambda: ctx.glv_mul_witness_batch(px, py, k, cols=cols[:, :n], ld=ld))
        b = n * (_p.GLV_MUL_COLS * 8 + 96)
        print(json.dumps({"kernel": "glv_mul pipeline", "n": n, "ms": round(ms, 4), "fills_per_s": round(n / ms * 1e3, 1),
                          "alg_bytes": b, "GBps": round(b / ms / 1e6, 1), "frac_hbm_peak": round(b / ms / 1e6 / PEAK, 4)}), flush=True)
        del cols
if "aux" in which:
    # built-in-generator columns from a finished 2^16 verify witness matrix.
    # algorithmic bytes per signature: 8 959 columns written + the columns each item reads (4 x 9 limbs of the two
    # selects, 1-2 digit limbs; MSM digits also gather 18 table limbs) = (66*38 + 73*58 + 60) * 8 B read
    n = 1 << 16
    sig = p2e.synth_signatures(seed=4, n=n)
    dev = [torch.from_numpy(a).cuda() for a in sig]
    ld = n + 16
    cols = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
    ctx.ecdsa_verify_witness_batch(*dev, cols=cols[:, :n], ld=ld)
    aux = torch.empty((p2e.VERIFY_AUX_COLS, ld), dtype=torch.int64, device="cuda")
    err = torch.empty(n, dtype=torch.uint8, device="cuda")
    ms = timed(lambda: ctx.aux_witness_batch(0, dev[4], cols, n=n, ld=ld, aux=aux[:, :n], err=err, ld_aux=ld))
    b = n * 8 * (p2e.VERIFY_AUX_COLS + 66 * 38 + 73 * 58 + 60)
    print(json.dumps({"kernel": "k_aux", "n": n, "aux_cols": p2e.VERIFY_AUX_COLS, "ms": round(ms, 4), "alg_bytes": b,
                      "GBps": round(b / ms / 1e6, 1), "frac_hbm_peak": round(b / ms / 1e6 / PEAK, 4)}), flush=True)
if "compact" in which:
    # compact transfer container: 661 KB read + 474 KB written per signature
    n = 1 << 15
    _m, nn, nw = p2e.compact_layout(0)
    ld = n + 16
    cols = torch.randint(0, 1 << 29, (p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
    nar = torch.empty((nn, ld), dtype=torch.int32, device="cuda")
    wid = torch.empty((nw, ld), dtype=torch.int64, device="cuda")
    err = torch.empty(n, dtype=torch.uint8, device="cuda")
    ms = timed(lambda: ctx.columns_compact(0, cols, n=n, ld=ld, narrow=nar, wide=wid, err=err, ld_narrow=ld, ld_wide=ld))
    b = n * (p2e.VERIFY_COLS * 8 + nn * 4 + nw * 8)
    print(json.dumps({"kernel": "k_compact", "n": n, "ms": round(ms, 4), "alg_bytes": b, "GBps": round(b / ms / 1e6, 1),
                      "frac_hbm_peak": round(b / ms / 1e6 / PEAK, 4)}), flush=True)
if "verdict" in which:
    # p2e_ecdsa_verify_batch: the verdict alone (scalar phase + Jacobian chains, no witness), the pre-filter of SURVEY 8(f) rank 4
    n = 1 << 16
    sig = p2e.synth_signatures(seed=4, n=n)
    dev = [torch.from_numpy(a).cuda() for a in sig]
    err = torch.empty(n, dtype=torch.uint8, device="cuda")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    ms = timed(lambda: ctx.ecdsa_verify_batch(*dev, err=err, valid=valid))
    assert int(valid.sum()) == n
    print(json.dumps({"kernel": "ecdsa_verify_batch (verdict only)", "n": n, "ms": round(ms, 4),
                      "verifies_per_s": round(n / ms * 1e3, 1)}), flush=True)
