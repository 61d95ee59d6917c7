#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for a in "150 40 1" "150 40 48" "158 40 4" "97 40 4" "90 40 4" "60 40 4" "65 40 48" "200 40 4"; do timeout 120 scripts/ubench/dense_probe $a; done
python scripts/bench_dense.py 2>&1 | tail -3
timeout 600 python -m pytest tests/test_flexible.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/bench_flex.py 16 2>&1 | tail -1
