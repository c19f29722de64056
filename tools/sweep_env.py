"""Sweep P2E_* tuning env vars in ONE process (fresh context per setting), interleaved rounds."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import plonky2_ecdsa_amd as p2e
n = 1 << 16
sigs = p2e.synth_signatures(seed=4, n=n)
dev = [torch.from_numpy(a).cuda() for a in sigs]
ld = n + 16
big = torch.empty((p2e.VERIFY_COLS, ld), dtype=torch.int64, device="cuda")
err = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
settings = []
for msm, fx in itertools.product((1, 2, 4, 6, 8, 10, 12), (1, 2, 3)):
    settings.append({"P2E_MSM_PIECES": str(msm), "P2E_FIXED_PIECES": str(fx)})
ctxs = []
for s in settings:
    os.environ.update(s)
    ctxs.append(p2e.Context(device=0))
res = [[] for _ in settings]
def step(c):
    c.ecdsa_verify_witness_batch(*dev, cols=big[:, :n], err=err, valid=valid, ld=ld)
for c in ctxs:
    step(c)
torch.cuda.synchronize()
for rnd in range(3):
    for k, c in enumerate(ctxs):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(4): step(c)
        torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / 4 * 1e3)
order = sorted(range(len(settings)), key=lambda k: sorted(res[k])[1])
for k in order[:12]:
    print({a[4:]: b for a, b in settings[k].items()}, "median %.3f ms" % sorted(res[k])[1])
print("worst", {a[4:]: b for a, b in settings[order[-1]].items()}, "median %.3f ms" % sorted(res[order[-1]])[1])
