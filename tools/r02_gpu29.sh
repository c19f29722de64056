#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r02.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_gpu_r02.log; tail -4 gpurun_out/pytest_gpu_r02.log
grep -q "pytest_exit=0" gpurun_out/pytest_gpu_r02.log || exit 1
{
for n in 8192 12288 14336 16384 24576 32768 40960 48896 65536; do python tools/mid_sweep.py $n; done
for n in 1024 16384 32768; do AB=1 python - <<PY
import sys, time, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import plonky2_ecdsa_amd as p2e
n = $n
sig = [torch.from_numpy(a).cuda() for a in p2e.synth_signatures(seed=3, n=n)]
ctx = p2e.Context(device=0)
cols = torch.empty((p2e.GLV_MUL_COLS, n + 16), dtype=torch.int64, device="cuda")
call = lambda: ctx.glv_mul_witness_batch(sig[3], sig[4], sig[0], cols=cols[:, :n], ld=n + 16)
for _ in range(3): call()
torch.cuda.synchronize(); ts = []
for _ in range(15):
    t = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort(); print(f"glv_mul n={n}: median {ts[7]:.3f} ms")
PY
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/mid_sweep3.log
cat gpurun_out/mid_sweep3.log
for lg in 13 14 15; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-log2 $lg --no-limb-split --no-p256 --no-cpu-baseline > gpurun_out/proxy_$lg.json 2> gpurun_out/proxy_$lg.err
  python -c "
import json; d=json.load(open('gpurun_out/proxy_$lg.json')); print('2^$lg:', d['value'], d['ms_per_step'], d['median_step_ms'], d.get('checked_vs_oracle'))"
done
