#!/bin/bash
# isolated crossing with the responses out: slab sizes / forms, same box; bits must agree
cd ${GRAFT_REPO_ROOT:-.}
run() { TAG="$1" python scripts/iso_xi.py 2>&1 | tail -1; }
RAFTX_XI_SLAB_PAIRS=0 run "two blocks, whole downloads"
RAFTX_XI_SLAB_PAIRS=0 RAFTX_XI_SLABS=1 run "round-4 first form (four blocks)"
for n in 1 2 3; do for p in 512 768 1024 1536; do RAFTX_XI_SLAB_STREAMS=$n RAFTX_XI_SLAB_PAIRS=$p run "slabs of $p on $n streams"; done; done
RAFTX_SWEEP_SPLIT=0.3,0.7 run "default slabs, blocks 30/70"
RAFTX_SWEEP_SPLIT=0.12,0.88 run "default slabs, blocks 12/88"
RAFTX_SWEEP_SPLIT=1 run "default slabs, one block"
