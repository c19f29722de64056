#!/bin/bash
for f in 1 2; do for p in 1 2 3 4 6; do
  echo -n "fixed=$f msm=$p  "; P2E_FIXED_PIECES=$f P2E_MSM_PIECES=$p timeout -k 10 200 python bench.py --no-cpu-baseline --steps 8 --warmup 2 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['roofline']['achieved'])"
done; done
