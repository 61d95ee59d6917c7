#!/usr/bin/env python
"""Kernel times of the geometry chain alone (nothing else on the GPU): python scripts/exp_member.py lib1.so ... under
rocprofv3 --kernel-trace --stats.  Tuning experiments only (variants may skip work)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from raft_amd._abi import RaftxLib             # noqa: E402
from raft_amd.backend import HIP_LIB_PATH      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "head"
path = HIP_LIB_PATH if name == "head" else os.path.join(ROOT, name)
ctx = RaftxLib(path).context(0)
sw, _, _ = bench.make_sweep(ctx, 10000, 0, pinned=False)
for i in range(6):
    sw.upload(ctx)
    ctx.synchronize()
ctx.close()
