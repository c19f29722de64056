cd "$GRAFT_REPO_ROOT"
python tools/dbg_quad_runs.py
P2E_RUNS_MIN_N=0 python tools/dbg_quad_runs.py
P2E_RUNS_MIN_N=0 P2E_QUAD_MAX_N=0 python tools/dbg_quad_runs.py
P2E_RUNS_MIN_N=0 P2E_BINV_SPLIT_LOG2=0 python tools/dbg_quad_runs.py
P2E_RUNS_MIN_N=0 P2E_MSM_PIECES_SMALL=2 python tools/dbg_quad_runs.py
P2E_RUNS_MIN_N=0 DBG_N=300 python tools/dbg_quad_runs.py
