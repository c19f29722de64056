"""count sporadic long calls: python stall_hunt.py lib.so ... (env AB_N, AB_PROG, ITERS)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import plonky2_ecdsa_amd as p2e
n = int(os.environ.get("AB_N", 1024)); prog = int(os.environ.get("AB_PROG", 1)); iters = int(os.environ.get("ITERS", 60))
print("HWQ", os.environ.get("GPU_MAX_HW_QUEUES"), "n", n, "prog", prog)
sigs = p2e.synth_signatures(seed=4, n=n)
dev = [torch.from_numpy(a).cuda() for a in sigs]
ld = n + 16
big = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path)); h = C.c_void_p()
    L.p2e_ecdsa_verify_witness_batch.restype = C.c_long; L.p2e_glv_mul_witness_batch.restype = C.c_long
    assert L.p2e_ctx_create(C.c_int(0), C.c_uint(0), C.c_void_p(0), C.byref(h)) == 0
    P = lambda t: C.c_void_p(t.data_ptr())
    def step():
        if prog == 1:
            rc = L.p2e_glv_mul_witness_batch(h, P(dev[3]), P(dev[4]), P(dev[0]), P(big), C.c_size_t(n), C.c_size_t(ld), P(err), P(valid))
        else:
            rc = L.p2e_ecdsa_verify_witness_batch(h, *[P(d) for d in dev], P(big), C.c_size_t(n), C.c_size_t(ld), P(err), P(valid))
        assert rc == 0
    step(); ts = []
    for _ in range(iters):
        t = time.perf_counter(); step(); ts.append((time.perf_counter() - t) * 1e3)
    a = np.array(ts); med = np.median(a)
    print(f"{os.path.basename(path):24s} median {med:.2f} max {a.max():.1f} outliers(>2x) {(a > 2 * med).sum()} at {np.nonzero(a > 2 * med)[0].tolist()[:8]}")
