#!/bin/bash
# ONE parameterised lease script (round 3 on; the 38 one-off r02_gpuN.sh files it replaces are described in
# tools/README.md and their logs are kept under profiles/).  Usage on the GPU box, through gpurun:
#   gpurun --timeout T -- 'tools/gpu_job.sh TAG job [job ...]'
# Every job writes gpurun_out/TAG_<job>.log (+ artefacts) and the script stops at the first job that fails or times
# out (no GPU step is started after a killed one).  Jobs:
#   pytest[:EXPR]     python -m pytest tests -m gpu [-k EXPR]
#   bench[:ARGS]      python bench.py ARGS            (ARGS with '+' for spaces, e.g. bench:--steps+20)
#   smoke             __graft_entry__.smoke()
#   profile           tools/profile_gpu.sh TAG: rocprofv3 --kernel-trace --stats of bench.py, then the PMC passes of the guide
#                     (FETCH_SIZE / WRITE_SIZE, separate runs) and the k_split calibration; tools/summarize_prof.py TAG
#                     condenses gpurun_out/prof_TAG into profiles/
#   timeline:P        rocprofv3 kernel timeline of one 2^16 step with P MSM pieces (tools/trace_timeline.sh)
#   py:SCRIPT[:ARGS]  python tools/SCRIPT ARGS
#   env:NAME=VALUE    export for the jobs that follow
#   ranks:N[:ARGS]    bench.py ARGS on N gloo ranks sharing the one GPU (rehearsal of the N>1 code paths)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # run NAME TIMEOUT cmd...
    local name=$1 t=$2; shift 2
    echo "== $TAG $name: $*"
    timeout -k 10 "$t" "$@" > "gpurun_out/${TAG}_${name}.log" 2>&1
    local rc=$?
    echo "exit=$rc" >> "gpurun_out/${TAG}_${name}.log"
    tail -n 6 "gpurun_out/${TAG}_${name}.log"
    return $rc
}
for job in "$@"; do
    kind=${job%%:*}; arg=""; [[ "$job" == *:* ]] && arg=${job#*:}
    case $kind in
    env)      export "$arg"; echo "== export $arg" ;;
    pytest)   if [ -n "$arg" ]; then run "pytest_${arg//[^A-Za-z0-9_]/_}" 1100 python -m pytest tests -m gpu -x -q -k "$arg" --durations=15
              else run pytest 1100 python -m pytest tests -m gpu -x -q --durations=25; fi || exit 1 ;;
    bench)    run "bench${arg:+_}${arg//[^A-Za-z0-9_]/_}" 900 python bench.py ${arg//+/ } || exit 1
              grep '^{' "gpurun_out/${TAG}_bench${arg:+_}${arg//[^A-Za-z0-9_]/_}.log" > "gpurun_out/${TAG}_bench${arg:+_}${arg//[^A-Za-z0-9_]/_}.json" ;;
    smoke)    run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1 ;;
    profile)  run profile 1100 tools/profile_gpu.sh "$TAG" || exit 1 ;;   # -> gpurun_out/prof_TAG; then tools/summarize_prof.py TAG here
    timeline) run "timeline_$arg" 400 tools/trace_timeline.sh "$arg" 2 || exit 1 ;;
    py)       script=${arg%%:*}; sargs=""; [[ "$arg" == *:* ]] && sargs=${arg#*:}
              run "py_${script%.py}${sargs:+_}${sargs//[^A-Za-z0-9_]/_}" 1100 python "tools/$script" ${sargs//+/ } || exit 1 ;;
    ranks)    # ranks:N[:ARGS]  bench.py on N gloo ranks sharing the one GPU (rehearsal of the N>1 code paths; ARGS as for bench)
              nr=${arg%%:*}; rargs=""; [[ "$arg" == *:* ]] && rargs=${arg#*:}
              run "ranks_${nr}${rargs:+_}${rargs//[^A-Za-z0-9_]/_}" 900 env P2E_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node "$nr" --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus "$nr" --steps 4 --warmup 1 --check 64 ${rargs//+/ } || exit 1 ;;
    *)        echo "unknown job $job"; exit 2 ;;
    esac
done
echo "== $TAG done"
