#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CP_PASSES=1 timeout -k 10 500 python tools/bench_curve_programs.py 14 > gpurun_out/curve_passes.jsonl 2> gpurun_out/curve_passes.err; echo "exit=$?"; grep passes gpurun_out/curve_passes.jsonl; tail -2 gpurun_out/curve_passes.err
