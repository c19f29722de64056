#!/bin/bash
# north star's "LDS-staged precomputed window tables": fixed-base table rows staged in LDS (k_chains: the piece's
# windows, k_expand: the op's 16 entries) against the default L2-resident gather.  Parity of the experimental build,
# same-process A/B of the whole step, rocprofv3 kernel stats of both builds.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(readlink -f "$(which python3)")
O=gpurun_out/r02_lds; mkdir -p $O
L=plonky2-ecdsa_amd/libp2e_hip.so; X=gpurun_exp_ldsfb.so
P2E_LIB=$PWD/$X P2E_QUAD_MAX_N=0 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "golden or ragged or edge" 2>&1 | tail -2
AB_N=65536 python tools/ab_libs.py "$L" "$X" 2>&1 | grep median | tee $O/ab_65536.txt
AB_N=8192 python tools/ab_libs.py "$L#P2E_QUAD_MAX_N=0" "$X#P2E_QUAD_MAX_N=0" 2>&1 | grep median | tee $O/ab_8192_lane.txt
for v in default lds; do
  LIBV=$L; [ $v = lds ] && LIBV=$X
  rm -rf $O/kt_$v
  P2E_LIB=$PWD/$LIBV rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- $PY bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-limb-split --check 0 > $O/bench_$v.json 2> $O/kt_$v.err
  cp $O/kt_$v/kt_kernel_stats.csv $O/kernel_stats_$v.csv; rm -rf $O/kt_$v
  head -12 $O/kernel_stats_$v.csv | cut -c1-150
done
