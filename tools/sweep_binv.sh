#!/bin/bash
for b in 64 32 22 15 11; do for p in 4 6 8; do
  echo -n "binv=$b msm=$p  "; P2E_BINV_TARGET=$b P2E_MSM_PIECES=$p timeout -k 10 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['roofline']['achieved'])"
done; done
