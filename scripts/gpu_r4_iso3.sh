#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for V in "A=1" "RAFTX_XI_SLABS=0" "RAFTX_SWEEP_SPLIT=0.2,0.3,0.3,0.2" "RAFTX_SWEEP_SPLIT=0.2,0.2,0.2,0.2,0.2" "A=2" "RAFTX_SWEEP_SPLIT=0.2,0.3,0.3,0.2"; do
  ( export $V; timeout 200 python scripts/iso_xi.py 2>&1 | tail -4 | awk '{print $4}' | tr '\n' ' '; echo " <- $V" )
done
