#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 600 python -m pytest tests/test_flexible.py -m gpu -x -q 2>&1 | tail -3 )
python scripts/bench_flex.py 8 2>&1 | tail -1
python scripts/bench_flex.py 64 2>&1 | tail -1
