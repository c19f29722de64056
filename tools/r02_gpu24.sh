#!/bin/bash
# one-GPU proxies of the strong-scaled shape (8 192 / 16 384 / 32 768 signatures per GPU) with the final library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for lg in 13 14 15; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch-log2 $lg --no-limb-split --no-p256 --no-cpu-baseline > gpurun_out/proxy_$lg.json 2> gpurun_out/proxy_$lg.err
  python -c "
import json; d=json.load(open('gpurun_out/proxy_$lg.json')); print('2^$lg:', d['value'], d['ms_per_step'], d['median_step_ms'], d.get('checked_vs_oracle'))"
done
