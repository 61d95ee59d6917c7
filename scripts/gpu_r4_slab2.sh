#!/bin/bash
# isolated crossing with the responses out, slabs of one residency round: block splits, same box
cd ${GRAFT_REPO_ROOT:-.}
run() { TAG="$1" python scripts/iso_xi.py 2>&1 | tail -1; }
run "default (20/80)"
for sp in 0.2,0.3,0.5 0.1,0.3,0.6 0.2,0.2,0.6 0.1,0.2,0.3,0.4 0.2,0.2,0.3,0.3 0.1,0.1,0.2,0.3,0.3 0.1,0.9 0.3,0.7; do RAFTX_SWEEP_SPLIT=$sp run "blocks $sp"; done
run "default (20/80) again"
