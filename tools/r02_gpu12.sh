#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python tools/bench_curve_programs.py 13 15 16 > gpurun_out/curve_programs.jsonl 2> gpurun_out/curve_programs.err; echo "exit=$?"; cat gpurun_out/curve_programs.jsonl; tail -3 gpurun_out/curve_programs.err
