#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 32768 8192; do for rep in 1 2 3 4 5 6; do python tools/mid_sweep.py $n 2>&1 | grep "^n=" | sed 's/valid.*//'; done; done
