import sys, time, numpy as np, cProfile, pstats
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd._abi import RaftxLib
from raft_amd import dropin, snapshot
from raft_amd import backend; ctx=backend.hip_library().context(0)
fx=snapshot.load_fixture('flex_volturnus.npz'); model=snapshot.build_model(fx['model'])
eng=dropin.Engine(ctx)
case=snapshot.case_from_fixture(fx['cases'][0])
eng.solveDynamics(model, dict(case))
pr=cProfile.Profile(); pr.enable()
for _ in range(3): eng.solveDynamics(model, dict(case))
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
