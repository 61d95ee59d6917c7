#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python scripts/bench_dense.py 2>&1 | tail -3
RAFTX_DENSE_L2=1 python scripts/bench_dense.py 2>&1 | tail -3
