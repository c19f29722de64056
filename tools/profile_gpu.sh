#!/bin/bash
# rocprofv3 evidence for one round: kernel trace + stats of bench.py, then PMC passes (separate runs,
# one counter family each, as MI355X_MICROARCH.md prescribes).  Usage: tools/profile_gpu.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
PY=$(readlink -f "$(which python3)")
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $PY bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_kt.json 2> $OUT/kt.err || echo "kt failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $PY bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-limb-split --check 0 > $OUT/bench_under_fetch.json 2> $OUT/fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $PY bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-limb-split --check 0 > $OUT/bench_under_write.json 2> $OUT/write.err || echo "write failed"
# calibration of the counters on a kernel with a known byte count (limb split: 32 B in / 72 B out per element)
REPS=2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cal -- $PY tools/bench_kernels.py split > $OUT/cal_fetch.json 2> $OUT/cal_fetch.err || echo "cal fetch failed"
REPS=2 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cal -- $PY tools/bench_kernels.py split > $OUT/cal_write.json 2> $OUT/cal_write.err || echo "cal write failed"
find $OUT -name '*.csv' | head -50
du -sh $OUT
