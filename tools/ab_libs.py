"""A/B two builds of libp2e_hip.so in ONE process, interleaved rounds (cdna guide rule 24).
env AB_N (default 65536) = batch, AB_PROG = 0 verify (default) / 1 glv_mul."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import plonky2_ecdsa_amd as p2e
libs = sys.argv[1:]
n = int(os.environ.get("AB_N", 1 << 16))
prog = int(os.environ.get("AB_PROG", 0))
sigs = p2e.synth_signatures(seed=4, n=n)
dev = [torch.from_numpy(a).cuda() for a in sigs]
ld = n + 16
big = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
ctxs = []
for spec in libs:   # "path", "path@R", "path@R@P" (R = P2E_RUN_ITERS, P = P2E_MSM_PIECES) and/or "...#K=V;K=V" (any P2E_* knobs)
    spec, _, extra = spec.partition("#")
    path, r, pcs = (spec.split("@") + ["", ""])[:3]
    for k in [k for k in os.environ if k.startswith("P2E_")]:
        os.environ.pop(k)
    for k, v in (("P2E_RUN_ITERS", r), ("P2E_MSM_PIECES", pcs)):
        if v:
            os.environ[k] = v
    for kv in filter(None, extra.split(";")):
        k, _, v = kv.partition("=")
        os.environ[k] = v
    L = C.CDLL(os.path.abspath(path))
    h = C.c_void_p()
    L.p2e_ecdsa_verify_witness_batch.restype = C.c_long
    L.p2e_glv_mul_witness_batch.restype = C.c_long
    assert L.p2e_ctx_create(C.c_int(0), C.c_uint(0), C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(h)) == 0
    ctxs.append((L, h))
os.environ.update({k: "" for k in ()})
def step(L, h):
    if prog == 1:
        rc = L.p2e_glv_mul_witness_batch(h, C.c_void_p(dev[3].data_ptr()), C.c_void_p(dev[4].data_ptr()), C.c_void_p(dev[0].data_ptr()),
                                         C.c_void_p(big.data_ptr()), C.c_size_t(n), C.c_size_t(ld), C.c_void_p(err.data_ptr()),
                                         C.c_void_p(valid.data_ptr()))
        assert rc >= 0, rc
        return
    rc = L.p2e_ecdsa_verify_witness_batch(h, *[C.c_void_p(d.data_ptr()) for d in dev], C.c_void_p(big.data_ptr()), C.c_size_t(n), C.c_size_t(ld),
                                          C.c_void_p(err.data_ptr()), C.c_void_p(valid.data_ptr()))
    assert rc >= 0, rc
res = {p: [] for p in libs}
for L, h in ctxs:
    step(L, h); step(L, h)
torch.cuda.synchronize()
for rnd in range(6):
    for p, (L, h) in zip(libs, ctxs):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): step(L, h)
        torch.cuda.synchronize(); res[p].append((time.perf_counter() - t) / 5 * 1e3)
for p in libs:
    r = sorted(res[p]); print(f"{os.path.basename(p):60s} median {r[len(r)//2]:.3f} ms  min {r[0]:.3f}  all {[round(x,2) for x in res[p]]}")
