import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from raft_amd import dropin, backend
from raft_amd._abi import RaftxLib
from tests.util import load_model_fixture, case_from_fixture, rel_err, group_rel_err
name = sys.argv[1] if len(sys.argv) > 1 else "c1_oc3spar.npz"
fx, model = load_model_fixture(name)
fx2, model2 = load_model_fixture(name)
hip = backend.default_context(0)
orc = RaftxLib("oracle/libraftx_oracle.so").context(0)
c = fx["cases"][0]
Xh = dropin.Engine(hip).solveDynamics(model, case_from_fixture(c)).copy()
Xo = dropin.Engine(orc).solveDynamics(model2, case_from_fixture(c)).copy()
f, g = model.fowtList[0], model2.fowtList[0]
print("niter", model._raftx_niter, model2._raftx_niter, "flags", model._raftx_flags, model2._raftx_flags)
print("Xi", group_rel_err(Xh[:1], Xo[:1]))
print("Z", rel_err(f.Z, g.Z), "F_iner", rel_err(f.F_hydro_iner, g.F_hydro_iner), "B", rel_err(f.B_hydro_drag, g.B_hydro_drag))
print("Fdrag", rel_err(f._raftx_Fdrag, g._raftx_Fdrag))
np.set_printoptions(linewidth=200, precision=4)
print(np.abs(Xh[0]-Xo[0]).max(axis=0)[:12], np.abs(Xo[0]).max(axis=0)[:12])
print(f.B_hydro_drag[0], g.B_hydro_drag[0])
