"""BASELINE config 5 shape on ONE GPU: witness columns of a large batch streamed to host memory in chunks
(plonky2_ecdsa_amd.stream.HostStreamer; the full batch does not fit anywhere: 2^20 signatures = 693 GB).
Prints one JSON line: PCIe-inclusive fills/s and D2H GB/s (never the bench.py value).  --check K compares K
signatures of EVERY chunk with the oracle on the host copy."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import plonky2_ecdsa_amd as p2e
from plonky2_ecdsa_amd.stream import HostStreamer

ap = argparse.ArgumentParser()
ap.add_argument("--total-log2", type=int, default=17)
ap.add_argument("--chunk-log2", type=int, default=13)
ap.add_argument("--rows", action="store_true", help="one contiguous 82615-element row per signature (transposed on the GPU)")
ap.add_argument("--compact", action="store_true", help="the compact container: 474 KB instead of 661 KB per signature over the link")
ap.add_argument("--check", type=int, default=2, help="signatures of EVERY chunk compared with the oracle on the host copy")
args = ap.parse_args()
total, chunk = 1 << args.total_log2, 1 << args.chunk_log2
container = "rows" if args.rows else "compact" if args.compact else "u64"
hs = HostStreamer(device=0, chunk=chunk, container=container)
sigs = p2e.synth_signatures(seed=5, n=total)
dev_in = [torch.from_numpy(a).cuda() for a in sigs]
mismatch = []
if args.check:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c


def consumer(ch):
    if not args.check:
        return
    idx = np.unique(np.linspace(0, ch.n - 1, args.check).astype(np.int64))
    want, _, _ = oracle_c.verify_witness(*[a[ch.first + idx] for a in sigs])
    if container == "compact":
        got = p2e.compact_expand(0, ch.narrow[:, idx].contiguous().numpy().view(np.uint32), ch.wide[:, idx].contiguous().numpy())
    elif container == "rows":
        got = ch.rows[idx].t().contiguous().numpy().view(np.uint64)
    else:
        got = ch.cols[:, idx].contiguous().numpy().view(np.uint64)
    if not np.array_equal(got, want):
        mismatch.append(ch.index)


st = hs.run(dev_in, consumer)
print(json.dumps({"workload": f"2^{args.total_log2} verifies streamed to pinned host memory in 2^{args.chunk_log2}-signature chunks, "
                              f"container {container}", "seconds": round(st["seconds"], 3),
                  "fills_per_s_pcie_inclusive": round(st["fills_per_s_pcie_inclusive"], 1), "d2h_GBps": round(st["d2h_GBps"], 2),
                  "bytes_d2h": st["bytes_d2h"], "flagged": st["flagged"], "valid": st["valid"], "chunks": st["chunks"],
                  "checked_per_chunk": args.check, "host_copy_matches_oracle": (not mismatch) if args.check else None}))
