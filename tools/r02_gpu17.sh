#!/bin/bash
# issue priority of the expansion waves (build flag -DP2E_EXPAND_PRIO=k): same-process A/B against the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
  timeout -k 10 300 python tools/ab_libs.py plonky2-ecdsa_amd/libp2e_hip.so tools/ab_build/libp2e_prio1.so tools/ab_build/libp2e_prio2.so tools/ab_build/libp2e_prio3.so > gpurun_out/ab_prio_$rep.log 2>&1; tail -4 gpurun_out/ab_prio_$rep.log
done
