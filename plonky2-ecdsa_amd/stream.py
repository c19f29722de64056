"""BASELINE config 5: the witness columns of a batch too large to stay resident (2^20 verifies = 693 GB),
streamed to the host prover in chunks (what plonky2's prove_with_partition_witness consumes).

One asynchronous p2e context fills chunk k into one of two device buffers while a copy stream moves chunk k-1
into one of two PINNED host buffers and the caller's consumer reads chunk k-2's ... more precisely, per iteration:

    enqueue fill(k)  ->  enqueue D2H(k) behind it on the copy stream  ->  wait D2H(k-1)  ->  consumer(chunk k-1)

so the GPU computes chunk k while the link moves chunk k-1 and the host consumes chunk k-1 from the other
pinned buffer.  No p2e_sync per chunk: err / valid bytes travel with the chunk and are counted on the host.
The link (PCIe Gen5 x16, about 56 GB/s measured) is the bound, not the GPU: the compact container
(include/p2e.h, 474 KB instead of 661 KB per verify) is the faster way through it.
"""
from __future__ import annotations

import time

import numpy as np

from . import (PROGRAM_VERIFY, VERIFY_AUX_COLS, VERIFY_COLS, Context, CurveProgram, P2EError,  # noqa: F401  (package namespace)
               compact_layout)


class Chunk:
    """One finished chunk on the host.  Valid only inside the consumer call (the pinned buffers are reused)."""
    __slots__ = ("index", "first", "n", "cols", "rows", "narrow", "wide", "err", "valid")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class HostStreamer:
    """container: "u64" (cols: (82615, n) int64 host tensor, column-major over the chunk), "rows" (rows: (n, 82615),
    one contiguous witness per signature, transposed on the GPU first) or "compact" (narrow (num_narrow, n) int32 +
    wide (num_wide, n) int64).

    curve_program = (kind, curve, blind): stream a curve program's fill instead (the P-256 verifier,
    or a stand-alone scalar multiplication whose inputs are (px, py, k)); containers "u64" and "rows"."""

    def __init__(self, device: int = 0, chunk: int = 8192, container: str = "u64", curve_program=None):
        import torch
        if container not in ("u64", "rows", "compact"):
            raise P2EError("container must be u64, rows or compact")
        if curve_program is not None and container == "compact":
            raise P2EError("curve programs write the u64 matrix only")
        self.torch = torch
        self.device, self.chunk, self.container = device, int(chunk), container
        dev = f"cuda:{device}"
        self.compute = torch.cuda.Stream(device=dev)
        self.copy = torch.cuda.Stream(device=dev)
        self.ctx = Context(device=device, stream=self.compute.cuda_stream, asynchronous=True)
        self.prog = CurveProgram(self.ctx, *curve_program) if curve_program is not None else None
        self.ncols = self.prog.num_cols if self.prog is not None else VERIFY_COLS
        # n + 16: a power-of-two column stride camps on the same HBM channels (see Context.ecdsa_verify_witness_batch)
        self.ld = ld = self.chunk + 16
        two = range(2)
        if container == "compact":
            _m, self.nn, self.nw = compact_layout(PROGRAM_VERIFY)
            self.d_nar = [torch.empty((self.nn, ld), dtype=torch.int32, device=dev) for _ in two]
            self.d_wid = [torch.empty((self.nw, ld), dtype=torch.int64, device=dev) for _ in two]
            self.h_nar = [torch.empty((self.nn, ld), dtype=torch.int32, pin_memory=True) for _ in two]
            self.h_wid = [torch.empty((self.nw, ld), dtype=torch.int64, pin_memory=True) for _ in two]
            self.bytes_per_chunk = (self.nn * 4 + self.nw * 8) * ld
        else:
            nc = self.ncols
            self.d_cols = [torch.empty((nc, ld), dtype=torch.int64, device=dev) for _ in two]
            if container == "rows":
                self.d_rows = [torch.empty((self.chunk, nc), dtype=torch.int64, device=dev) for _ in two]
                self.h_rows = [torch.empty((self.chunk, nc), dtype=torch.int64, pin_memory=True) for _ in two]
                self.bytes_per_chunk = self.chunk * nc * 8
            else:
                self.h_cols = [torch.empty((nc, ld), dtype=torch.int64, pin_memory=True) for _ in two]
                self.bytes_per_chunk = nc * ld * 8
        self.d_err = [torch.empty(self.chunk, dtype=torch.uint8, device=dev) for _ in two]
        self.d_valid = [torch.empty(self.chunk, dtype=torch.uint8, device=dev) for _ in two]
        self.h_err = [torch.empty(self.chunk, dtype=torch.uint8, pin_memory=True) for _ in two]
        self.h_valid = [torch.empty(self.chunk, dtype=torch.uint8, pin_memory=True) for _ in two]
        self.done_compute = [torch.cuda.Event() for _ in two]
        self.done_copy = [torch.cuda.Event() for _ in two]

    def _enqueue(self, k, sl, n):
        torch, b = self.torch, k & 1
        self.compute.wait_event(self.done_copy[b])          # device buffer b is free once its previous copy is done
        with torch.cuda.stream(self.compute):
            if self.container == "compact":
                self.ctx.ecdsa_verify_witness_compact_batch(*sl, narrow=self.d_nar[b], wide=self.d_wid[b], err=self.d_err[b],
                                                            valid=self.d_valid[b], ld_narrow=self.ld, ld_wide=self.ld)
            else:
                if self.prog is None:
                    self.ctx.ecdsa_verify_witness_batch(*sl, cols=self.d_cols[b], err=self.d_err[b], valid=self.d_valid[b], ld=self.ld)
                elif len(sl) == 5:
                    self.prog.verify_witness_batch(*sl, cols=self.d_cols[b], err=self.d_err[b], valid=self.d_valid[b], ld=self.ld)
                else:
                    self.prog.mul_witness_batch(*sl, cols=self.d_cols[b], err=self.d_err[b], valid=self.d_valid[b], ld=self.ld)
                if self.container == "rows":
                    self.ctx.columns_to_rows(self.d_cols[b], n=n, ld=self.ld, rows=self.d_rows[b])
            self.done_compute[b].record(self.compute)
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(self.done_compute[b])
            if self.container == "compact":
                self.h_nar[b].copy_(self.d_nar[b], non_blocking=True)
                self.h_wid[b].copy_(self.d_wid[b], non_blocking=True)
            elif self.container == "rows":
                self.h_rows[b][:n].copy_(self.d_rows[b][:n], non_blocking=True)
            else:
                self.h_cols[b].copy_(self.d_cols[b], non_blocking=True)
            self.h_err[b].copy_(self.d_err[b], non_blocking=True)
            self.h_valid[b].copy_(self.d_valid[b], non_blocking=True)
            self.done_copy[b].record(self.copy)

    def _chunk(self, k, first, n):
        b = k & 1
        kw = dict(index=k, first=first, n=n, err=self.h_err[b][:n].numpy(), valid=self.h_valid[b][:n].numpy())
        if self.container == "compact":
            kw.update(narrow=self.h_nar[b][:, :n], wide=self.h_wid[b][:, :n])
        elif self.container == "rows":
            kw.update(rows=self.h_rows[b][:n])
        else:
            kw.update(cols=self.h_cols[b][:, :n])
        return Chunk(**kw)

    def run(self, dev_inputs, consumer=None):
        """dev_inputs: the five (total, 32) uint8 DEVICE tensors (msg, r, s, pk.x, pk.y) of the whole stream (160 B per
        signature: they stay resident); (px, py, k) for a multiplication program.  consumer(chunk) is called once per chunk, in order, on the host copy.
        Returns {"seconds", "fills_per_s_pcie_inclusive", "d2h_GBps", "bytes_d2h", "flagged", "valid", "chunks"}."""
        torch = self.torch
        total = int(dev_inputs[0].shape[0])
        nchunks = -(-total // self.chunk)
        spans = [(k * self.chunk, min(self.chunk, total - k * self.chunk)) for k in range(nchunks)]
        flagged = valid = 0
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        for k in range(nchunks + 1):
            if k < nchunks:
                first, n = spans[k]
                self._enqueue(k, [d[first:first + n] for d in dev_inputs], n)
            if k >= 1:
                first, n = spans[k - 1]
                self.done_copy[(k - 1) & 1].synchronize()           # chunk k-1 is on the host; chunk k is computing
                ch = self._chunk(k - 1, first, n)
                flagged += int(np.count_nonzero(ch.err))
                valid += int(ch.valid.sum())
                if consumer is not None:
                    consumer(ch)
        torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - t0
        rc = self.ctx.sync()                                          # the asynchronous context's own status
        if rc < 0:
            raise P2EError("streamed fill failed")
        nbytes = nchunks * self.bytes_per_chunk
        return {"seconds": dt, "fills_per_s_pcie_inclusive": total / dt, "d2h_GBps": nbytes / dt / 1e9, "bytes_d2h": nbytes,
                "flagged": flagged, "valid": valid, "chunks": nchunks}
