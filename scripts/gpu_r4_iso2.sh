#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 600 python -m pytest tests/test_geometry.py -m gpu -x -q 2>&1 | tail -3 )
for V in "RAFTX_XI_SLABS=5" "RAFTX_XI_SLABS=4" "RAFTX_XI_SLABS=6"; do
  ( export $V; timeout 200 python scripts/iso_xi.py 2>&1 | tail -3 | tr '\n' ' '; echo " <- $V" )
done
