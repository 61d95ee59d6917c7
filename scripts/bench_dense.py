"""raftx_solve_dense kernel time (HIP events) on the flexible deck's shape: 150 DOFs, 1 right-hand side, 40 bins -- and at 200 bins."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raft_amd import backend
ctx = backend.default_context(0)
rng = np.random.default_rng(3)
for n, nw in ((150, 40), (150, 200), (60, 40)):
    w = np.linspace(0.1, 2.0, nw)
    M = rng.normal(size=(n, n)); M = M @ M.T + n * np.eye(n)
    B = rng.normal(size=(n, n)); B = B @ B.T
    C = rng.normal(size=(n, n)); C = C @ C.T + n * np.eye(n)
    F = rng.normal(size=(1, n, nw)) + 1j * rng.normal(size=(1, n, nw))
    ks = []
    for i in range(6):
        X = ctx.solve_dense(w, M, B, C, F)
        ks.append(ctx.last_kernel_ms())
    Z = -(w[None, None, :] ** 2) * M[:, :, None] + 1j * w[None, None, :] * B[:, :, None] + C[:, :, None]
    res = max(np.abs(Z[:, :, i] @ X[0, :, i] - F[0, :, i]).max() / np.abs(F[0, :, i]).max() for i in range(nw))
    print("n=%d nw=%d kernel ms %s  residual %.1e  (RAFTX_DENSE_L2=%s)" % (n, nw, ["%.3f" % k for k in ks[2:]], res, os.environ.get("RAFTX_DENSE_L2", "0")))
