"""LDS-race guard of SURVEY.md section 5: the responses must not depend on how the frequency bins are laid over lanes
and waves.  The launch shape of the fused kernel is fixed per process (RAFTX_SHAPE, read once), so every shape runs in
its own process: 1, 2, 4 and 8 waves per pair, 1 to 4 bins per lane.  Reductions group the bins differently per shape,
so agreement is to rounding (1e-12 of the RAOs), with identical iteration counts; two runs of one shape are bit-identical."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from raft_amd import backend, dropin, snapshot
fx = snapshot.load_fixture("c2_volturnus.npz")
model = snapshot.build_model(fx["model"])
eng = dropin.Engine(backend.default_context(0))
out = {}
for i, c in enumerate(fx["cases"]):
    case = {k: (list(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in c["case"].items()}
    out["Xi%%d" %% i] = eng.solveDynamics(model, case).copy()
    out["niter%%d" %% i] = np.array(model._raftx_niter)
np.savez(sys.argv[1], **out)
''' % ROOT


def run_shape(shape, path):
    env = dict(os.environ)
    if shape is None:
        env.pop("RAFTX_SHAPE", None)
    else:
        env["RAFTX_SHAPE"] = shape
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, env=env, timeout=300)
    return dict(np.load(path))


@pytest.mark.gpu
def test_responses_do_not_depend_on_the_launch_shape(tmp_path):
    from tests.util import group_rel_err
    base = run_shape(None, str(tmp_path / "default.npz"))
    again = run_shape(None, str(tmp_path / "again.npz"))
    for k in base:
        assert np.array_equal(base[k].view(np.uint8), again[k].view(np.uint8)), "repeat runs differ in " + k
    for shape in ("4,64", "2,128", "1,256", "2,256", "2,512"):        # 1, 2, 4, 4 and 8 waves per (design, sea state)
        r = run_shape(shape, str(tmp_path / ("s" + shape.replace(",", "_") + ".npz")))
        for k in base:
            if k.startswith("niter"):
                assert np.array_equal(base[k], r[k]), (shape, k)
            else:
                nH = base[k].shape[0] - 1
                assert group_rel_err(r[k][:nH], base[k][:nH]) < 1e-12, (shape, k)
