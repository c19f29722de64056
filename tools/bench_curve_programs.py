"""Curve programs: fills/s and the roofline of their expansion kernel.
One JSON line per (program, curve, batch): whole call (wall, median of REPS) and the expansion launches alone
(kc_expand: algorithmic bytes = columns x 8 B x n, durations = the library's own HIP event pairs around every launch,
p2e_last_phase_ms).  Usage: python tools/bench_curve_programs.py [log2 n ...]   (default 13 15)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import plonky2_ecdsa_amd as p2e

PEAK = 8000.0
reps = int(os.environ.get("REPS", "5"))
logs = [int(a) for a in sys.argv[1:]] or [13, 15]
ctx = p2e.Context(device=0, phase_timing=True)
# any point of the curve serves as the circuit's rand() point: 0xC0FFEE * G, computed by the synthetic-signature helper's
# own arithmetic is not exposed, so take a public key of the synthetic stream (a uniformly random multiple of G)
for curve, cname in ((p2e.CURVE_SECP256K1, "secp256k1"), (p2e.CURVE_P256, "p256")):
    bsig = p2e.synth_signatures_curve(curve, seed=777, n=1)
    blind = (int.from_bytes(bytes(bsig[3][0]), "little"), int.from_bytes(bytes(bsig[4][0]), "little"))
    for kind, kname in ((p2e.CP_WINDOWED_MUL, "curve_scalar_mul_windowed"), (p2e.CP_SCALAR_MUL, "curve_scalar_mul"),
                        (p2e.CP_VERIFY, "verify_p256_message_circuit")):
        if kind == p2e.CP_VERIFY and curve != p2e.CURVE_P256:
            continue
        prog = p2e.CurveProgram(ctx, kind, curve, blind)
        for lg in logs:
            n = 1 << lg
            sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures_curve(curve, seed=5, n=n)]
            ld = n + 16
            cols = torch.empty((prog.num_cols, ld), dtype=torch.int64, device="cuda")
            err = torch.empty(n, dtype=torch.uint8, device="cuda")
            valid = torch.empty(n, dtype=torch.uint8, device="cuda")
            if kind == p2e.CP_VERIFY:
                call = lambda: prog.verify_witness_batch(*sig, cols=cols[:, :n], err=err, valid=valid, ld=ld)
            else:
                call = lambda: prog.mul_witness_batch(sig[3], sig[4], sig[0], cols=cols[:, :n], err=err, valid=valid, ld=ld)
            _, _, _, bad = call()
            torch.cuda.synchronize()
            ok = int(valid.sum().item())
            ts, ph = [], []
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                call()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t) * 1e3)
                ph.append(ctx.last_phase_ms())
            ts.sort()
            ms = ts[len(ts) // 2]
            kinds = (("expand", "kc_expand"), ("runs", "kc_expand_runs"), ("fbrun", "kc_expand_fb_run"))
            p0 = ph[0]
            tot_ms = sorted(sum(p[k] for k, _ in kinds) for p in ph)[len(ph) // 2]
            tot_cols = sum(p0[k + "_cols"] for k, _ in kinds)
            out_bytes = prog.num_cols * 8 * n
            per_kernel = {}
            for k, kname2 in kinds:
                if p0[k + "_launches"]:
                    msk = sorted(p[k] for p in ph)[len(ph) // 2]
                    per_kernel[kname2] = {"launches": int(p0[k + "_launches"]), "cols_per_fill": int(p0[k + "_cols"]), "sum_launch_ms": round(msk, 3),
                                          "achieved_GBps": round(p0[k + "_cols"] * 8 * n / msk / 1e6, 1),
                                          "frac": round(p0[k + "_cols"] * 8 * n / msk / 1e6 / PEAK, 4)}
            alg = tot_cols * 8 * n
            print(json.dumps({"program": kname, "curve": cname, "n": n, "cols_per_fill": prog.num_cols, "flagged": bad, "valid": ok,
                              "ms": round(ms, 3), "fills_per_s": round(n / ms * 1e3, 1),
                              "whole_fill_GBps": round(out_bytes / ms / 1e6, 1), "whole_fill_frac_hbm_peak": round(out_bytes / ms / 1e6 / PEAK, 4),
                              "roofline": {"kernel": "expansion kernels (" + " + ".join(per_kernel) + ")", "bound": "hbm",
                                           "algorithmic_bytes": int(alg), "sum_launch_ms": round(tot_ms, 3),
                                           "achieved": round(alg / tot_ms / 1e6, 1), "peak": PEAK, "unit": "GB/s",
                                           "frac": round(alg / tot_ms / 1e6 / PEAK, 4), "per_kernel": per_kernel},
                              "scalar_ms": round(p0["scalar"], 3), "scratch_GB": round(prog.scratch_bytes(n) / 1e9, 2)}), flush=True)
            if os.environ.get("CP_PASSES") and lg <= 14:
                # the three streaming passes over the finished matrix (aux -> gate-internal -> U29 blocks): wall time of
                # each synchronous call, algorithmic bytes = what the pass must read + write
                inputs = sig if kind == p2e.CP_VERIFY else (sig[3], sig[4], sig[0])
                aux = torch.empty((prog.num_aux_cols, n), dtype=torch.int64, device="cuda")
                ux = torch.empty((prog.num_ux_cols, n), dtype=torch.int32, device="cuda")

                def wall(fn):
                    fn()
                    torch.cuda.synchronize()
                    ts2 = []
                    for _ in range(reps):
                        t = time.perf_counter()
                        fn()
                        torch.cuda.synchronize()
                        ts2.append((time.perf_counter() - t) * 1e3)
                    return sorted(ts2)[len(ts2) // 2]
                ms_aux = wall(lambda: prog.aux_witness_batch(inputs, cols[:, :n], n=n, ld=ld, aux=aux, err=err))
                ms_ux = wall(lambda: prog.ux_witness_batch(inputs, cols[:, :n], aux, n=n, ld=ld, ux=ux, err=err, u32=True))
                rec = {"program": kname, "curve": cname, "n": n, "passes": {
                    "kc_aux": {"cols": prog.num_aux_cols, "ms": round(ms_aux, 3), "written_GBps": round(prog.num_aux_cols * 8 * n / ms_aux / 1e6, 1)},
                    "kc_ux (u32)": {"cols": prog.num_ux_cols, "ms": round(ms_ux, 3), "written_GBps": round(prog.num_ux_cols * 4 * n / ms_ux / 1e6, 1),
                                    "frac_hbm_peak_written": round(prog.num_ux_cols * 4 * n / ms_ux / 1e6 / PEAK, 4)}}}
                if prog.num_gate_cols:
                    gate = torch.empty((prog.num_gate_cols, n), dtype=torch.int64, device="cuda")
                    ms_gate = wall(lambda: prog.gate_internal_batch(aux, n=n, gate=gate))
                    rec["passes"]["k_gate"] = {"cols": prog.num_gate_cols, "ms": round(ms_gate, 3),
                                               "written_GBps": round(prog.num_gate_cols * 8 * n / ms_gate / 1e6, 1)}
                    del gate
                print(json.dumps(rec), flush=True)
                del aux, ux
            del cols, sig
        prog.close()
