#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; echo "bench_exit=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_r02.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_k_expand']['frac'], d['whole_fill']['frac_of_hbm_peak']); print(d['p256_verify'])"
