"""Streaming passes beyond the fill at batch 2^K (default 16): built-in-generator columns (k_aux), constraint-block
columns (k_ux, u32 and u64) and the wire-matrix assembly (k_assemble, chunk of signatures), algorithmic GB/s each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import plonky2_ecdsa_amd as p2e
from plonky2_ecdsa_amd.wiremap import synthetic_wire_map
K = int(os.environ.get("K", 16)); n = 1 << K
sigs = p2e.synth_signatures(seed=4, n=n)
ctx = p2e.Context(device=0)
dev = [torch.from_numpy(a).cuda() for a in sigs]
cols, _e, valid, bad = ctx.ecdsa_verify_witness_batch(*dev)
aux, _, _ = ctx.aux_witness_batch(0, dev[4], cols)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2]
out = {"n": n}
for u32 in (True, False):
    ux = torch.empty((p2e.VERIFY_UX_COLS, n + 16), dtype=torch.int32 if u32 else torch.int64, device="cuda")
    dt = timed(lambda: ctx.ux_witness_batch(0, dev, cols, aux, ux=ux[:, :n]))
    written = p2e.VERIFY_UX_COLS * n * (4 if u32 else 8)
    read = 36000 * 8 * n            # ~ operand + result limbs read per signature (36 k limb loads)
    out["k_ux_u32" if u32 else "k_ux_u64"] = {"ms": round(dt * 1e3, 3), "GBps_written": round(written / dt / 1e9, 1), "bytes_written": written}
    if u32:
        assert int((ux[:, :n] >> 29).ne(0).sum()) == 0          # every U29 value in range, whole batch
        keep = ux
    else:
        del ux
src, dst, nw, deg = synthetic_wire_map(0)
wm = ctx.wire_map(0, src, dst, nw, deg)
chunk = min(n, 2048)
wires = torch.zeros((chunk, nw * deg), dtype=torch.int64, device="cuda")
dt = timed(lambda: ctx.assemble_wires(wm, cols[:, :chunk], aux[:, :chunk], keep[:, :chunk], wires=wires, n=chunk))
moved = len(src) * chunk
out["k_assemble"] = {"signatures": chunk, "entries": len(src), "ms": round(dt * 1e3, 3), "GBps_algorithmic_read_plus_write": round(moved * (8 + 8) / dt / 1e9, 1),
                     "Gvalues_per_s": round(moved / dt / 1e9, 2)}
print(json.dumps(out))
