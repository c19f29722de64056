#!/bin/bash
# one GPU call: full GPU test suite, headline bench, rocprofv3 evidence, kernel micro-benchmarks
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest_exit=$?" >> gpurun_out/pytest_gpu_$TAG.log; tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench_exit=$?"; cat gpurun_out/bench_$TAG.json
timeout -k 10 900 tools/profile_gpu.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -3 gpurun_out/profile_$TAG.log
timeout -k 10 200 tools/trace_timeline.sh 8 2 > gpurun_out/timeline_$TAG.txt 2>&1; tail -30 gpurun_out/timeline_$TAG.txt
timeout -k 10 200 python tools/bench_kernels.py > gpurun_out/kernels_$TAG.jsonl 2>&1; cat gpurun_out/kernels_$TAG.jsonl
timeout -k 10 300 python tools/stream_to_host.py --total-log2 17 > gpurun_out/stream_$TAG.json 2>&1; cat gpurun_out/stream_$TAG.json
