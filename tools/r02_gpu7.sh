#!/bin/bash
cd "$GRAFT_REPO_ROOT"
L=plonky2-ecdsa_amd/libp2e_hip.so
AB_N=65536 python tools/ab_libs.py "$L" "$L#P2E_EXPAND_LDS=54000" "$L#P2E_EXPAND_LDS=80000" "$L#P2E_EXPAND_LDS=160000" 2>&1 | grep median
AB_N=65536 python tools/ab_libs.py "$L" "$L#P2E_MSM_PIECES=10" "$L#P2E_FIXED_PIECES=3" "$L#P2E_EXPAND_LDS=80000;P2E_MSM_PIECES=10" 2>&1 | grep median
