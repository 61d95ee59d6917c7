"""A stand-in for one rank of bench.py (tests/test_bench_launcher.py): checks the environment bench.py's own launcher
hands to its ranks, meets the other ranks over the product's host transport, and lets rank 0 print the JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raft_amd import comm as rcomm          # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert len(os.environ["RAFTX_COMM_TOKEN"]) == 32 and os.environ["RAFTX_COMM_PORT"] != os.environ["MASTER_PORT"]
mode = os.environ.get("STUB_MODE", "ok")
if mode == "fail" and rank == 1:
    sys.exit(3)
if mode == "fail":
    time.sleep(60)                           # the launcher must stop us when rank 1 dies
    sys.exit(0)
c, kind = rcomm.from_env(None, prefer="host")
tot = c.gather_floats([float(rank + 1)])
c.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(tot.sum()), "argv": sys.argv[1:], "kind": kind}))
c.close()
