"""BASELINE config 5 shape on ONE GPU: witness columns of a large batch streamed to host memory in chunks
(the full batch does not fit anywhere: 2^20 signatures = 693 GB; one GPU's share of it, 2^17 = 86.6 GB).
Chunks of 2^13 signatures (5.4 GB of columns) are computed into two alternating device buffers and copied
D2H into two alternating PINNED host buffers on a copy stream, overlapping compute of chunk k+1 with the
copy of chunk k.  Prints one JSON line: PCIe-inclusive fills/s and D2H GB/s (never the bench.py value)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import plonky2_ecdsa_amd as p2e

ap = argparse.ArgumentParser()
ap.add_argument("--total-log2", type=int, default=17)
ap.add_argument("--chunk-log2", type=int, default=13)
ap.add_argument("--rows", action="store_true", help="transpose each chunk on the GPU so the host receives one contiguous "
                "82615-element row per signature (what a per-signature PartialWitness fill reads)")
ap.add_argument("--compact", action="store_true", help="compute each chunk straight into the compact container (u32 narrow + u64 wide "
                "matrices, p2e_ecdsa_verify_witness_compact_batch): 474 KB instead of 661 KB per signature over the host link")
ap.add_argument("--check", type=int, default=4, help="signatures of the LAST chunk to verify against the oracle on the host copy")
args = ap.parse_args()
total, chunk = 1 << args.total_log2, 1 << args.chunk_log2
nchunks = total // chunk
ld = chunk + 16
ctx = p2e.Context(device=0)
compute = torch.cuda.current_stream()
copy = torch.cuda.Stream()
dev_cols = [] if args.compact else [torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda") for _ in range(2)]
if args.rows:
    dev_rows = [torch.empty((chunk, p2e.VERIFY_COLS), dtype=torch.int64, device="cuda") for _ in range(2)]
    host_cols = [torch.empty((chunk, p2e.VERIFY_COLS), dtype=torch.int64, pin_memory=True) for _ in range(2)]
elif args.compact:
    _map, NN, NW = p2e.compact_layout(0)
    dev_nar = [torch.empty((NN, ld), dtype=torch.int32, device="cuda") for _ in range(2)]
    dev_wid = [torch.empty((NW, ld), dtype=torch.int64, device="cuda") for _ in range(2)]
    host_nar = [torch.empty((NN, ld), dtype=torch.int32, pin_memory=True) for _ in range(2)]
    host_wid = [torch.empty((NW, ld), dtype=torch.int64, pin_memory=True) for _ in range(2)]
else:
    host_cols = [torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, pin_memory=True) for _ in range(2)]
err = torch.empty(chunk, dtype=torch.uint8, device="cuda")
valid = torch.empty(chunk, dtype=torch.uint8, device="cuda")
# inputs for the whole stream stay on the device (160 B per signature)
sigs = p2e.synth_signatures(seed=5, n=total)
dev_in = [torch.from_numpy(a).cuda() for a in sigs]
done_compute = [torch.cuda.Event() for _ in range(2)]
done_copy = [torch.cuda.Event() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
bad_total = 0
for k in range(nchunks):
    b = k & 1
    compute.wait_event(done_copy[b])            # buffer b free again (its previous copy finished)
    sl = [d[k * chunk:(k + 1) * chunk] for d in dev_in]
    if args.compact:   # the fused schedule writes the compact container directly
        bad = ctx.ecdsa_verify_witness_compact_batch(*sl, narrow=dev_nar[b], wide=dev_wid[b], err=err, valid=valid,
                                                     ld_narrow=ld, ld_wide=ld)[4]
    else:
        _, _, _, bad = ctx.ecdsa_verify_witness_batch(*sl, cols=dev_cols[b][:, :chunk], err=err, valid=valid, ld=ld)
    bad_total += bad
    if args.rows:
        ctx.columns_to_rows(dev_cols[b], n=chunk, ld=ld, rows=dev_rows[b])
    done_compute[b].record(compute)
    with torch.cuda.stream(copy):
        copy.wait_event(done_compute[b])
        if args.compact:
            host_nar[b].copy_(dev_nar[b], non_blocking=True)
            host_wid[b].copy_(dev_wid[b], non_blocking=True)
        else:
            host_cols[b].copy_(dev_rows[b] if args.rows else dev_cols[b], non_blocking=True)
        done_copy[b].record(copy)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bytes_d2h = nchunks * p2e.VERIFY_COLS * (chunk if args.rows else ld) * 8
if args.compact:
    bytes_d2h = nchunks * (NN * 4 + NW * 8) * ld
ok = None
if args.check:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    last = (nchunks - 1) * chunk
    want, _, _ = oracle_c.verify_witness(*[a[last:last + args.check] for a in sigs])
    if args.compact:
        b = (nchunks - 1) & 1
        got = p2e.compact_expand(0, host_nar[b][:, :args.check].contiguous().numpy().view(np.uint32),
                                 host_wid[b][:, :args.check].contiguous().numpy())
    else:
        hb = host_cols[(nchunks - 1) & 1]
        got = (hb[:args.check].t() if args.rows else hb[:, :args.check]).contiguous().numpy().view(np.uint64)
    ok = bool(np.array_equal(got, want))
print(json.dumps({"workload": f"2^{args.total_log2} verifies streamed to pinned host memory in 2^{args.chunk_log2}-signature chunks"
                              + (", one contiguous row per signature" if args.rows else
                                 ", compact column-major chunks (u32 narrow + u64 wide)" if args.compact else ", column-major chunks"),
                  "seconds": round(dt, 3), "fills_per_s_pcie_inclusive": round(total / dt, 1),
                  "d2h_GBps": round(bytes_d2h / dt / 1e9, 2), "bytes_d2h": bytes_d2h, "flagged": bad_total,
                  "host_copy_matches_oracle": ok}))
