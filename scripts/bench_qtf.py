"""Config-5-shaped measurement of the slender-body QTF kernels: 200 x 200 second-order grid, nSet independent (heading,
motion) sets per launch -- on the VolturnUS-S strip table and on the BASELINE configs[4] deck itself (OC4semi-RAFT_QTF:
MacCamy-Fuchs columns with heave plates, inclined braces, Kim & Yue correction).  Prints one JSON line (kernel time from HIP events on the ctx
stream; the numpy oracle timed on a 40 x 40 sub-grid as the CPU datapoint)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import backend, qtf as rq, waves
from raft_amd import snapshot as standin

n_set = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fx = standin.load_fixture("refgold_qtf_VolturnUS-S.npz")
f = standin.build_model(fx["model"]).fowtList[0]
tab = rq.pack_qtf(f)
nw2 = 200
w2 = np.arange(1, nw2 + 1) * 0.0025 * 2 * np.pi
k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
rng = np.random.default_rng(0)
amp = np.array([1.0, 0.3, 0.7, 0.01, 0.02, 0.004])[:, None] / (1.0 + (w2[None, :] / 0.6) ** 2)
Xi = np.array([amp * np.exp(1j * (rng.uniform(0, 6, 6)[:, None] + 1.5 * w2[None, :])) for _ in range(n_set)])
betas = rng.uniform(0, 2 * np.pi, n_set)
Ms = np.array([f.M_struc] * n_set)
ctx = backend.default_context(0)
for _ in range(2):
    q = ctx.qtf_slender([tab] * n_set, Xi, betas, w2, k2, f.depth, f.rho_water, f.g, Ms)
ms = ctx.last_kernel_ms()
pairs = n_set * nw2 * (nw2 + 1) // 2
S = tab.strips.shape[0]
# CPU datapoint: numpy oracle on a 40-bin sub-grid
from oracle import qtf_oracle
sub = slice(0, nw2, 5)
t0 = time.perf_counter()
qtf_oracle.qtf_slender_body(tab, Xi[0][:, sub], betas[0], w2[sub], k2[sub], f.depth, f.rho_water, f.g, f.M_struc)
t_cpu = time.perf_counter() - t0
n_sub = len(w2[sub])
print(json.dumps({"metric": "QTF strip-pairs per second", "sets": n_set, "nw2": nw2, "strips": int(S),
                  "kernel_ms": ms, "strip_pairs_per_s": pairs * S / (ms * 1e-3),
                  "numpy_oracle_strip_pairs_per_s_1core": (n_sub * (n_sub + 1) // 2) * S / t_cpu,
                  "reference_strip_pairs_per_s_1core": 1.0 / 0.58e-3,
                  "hermitian_ok": bool(np.allclose(q[0], np.conj(np.transpose(q[0], (1, 0, 2))), atol=1e-6 * np.abs(q[0]).max()))}))

# ---- the configs[4] deck: OC4semi strip table (MacCamy-Fuchs + Kim & Yue), same 200 x 200 grid, Kim & Yue table + QTF launch
fxo = standin.load_fixture("c5_oc4semi_qtf.npz")
fo = standin.build_model(fxo["model"]).fowtList[0]
tabo = rq.pack_qtf(fo)
ko = np.array([waves.wave_number(x, fo.depth) for x in w2])
Mso = np.array([fo.M_struc] * n_set)
for _ in range(2):
    t0 = time.perf_counter()
    ctx.qtf_kay([tabo] * n_set, betas, w2, ko, fo.depth, fo.rho_water, fo.g)
    ms_k = ctx.last_kernel_ms()
    qo = ctx.qtf_slender([tabo] * n_set, Xi, betas, w2, ko, fo.depth, fo.rho_water, fo.g, Mso, None)
    ms_q = ctx.last_kernel_ms()
    wall = time.perf_counter() - t0
So = tabo.strips.shape[0]
print(json.dumps({"metric": "QTF strip-pairs per second, OC4semi-RAFT_QTF deck (configs[4])", "sets": n_set, "nw2": nw2, "strips": int(So),
                  "qtf_kernels_ms": ms_q, "kim_yue_kernels_ms": ms_k, "strip_pairs_per_s": pairs * So / (ms_q * 1e-3),
                  "wall_ms_incl_upload_and_download": 1e3 * wall,
                  "one_200x200_qtf_ms": (ms_q + ms_k) / n_set, "reference_one_200x200_qtf_s_1core": 20100 * So * 0.58e-3}))

# Kim & Yue correction table: OC4semi (MacCamy-Fuchs columns with heave plates), same 200 x 200 grid, device vs the
# host SciPy implementation it replaces on the batched path
fx5 = standin.load_fixture("c5_oc4semi_qtf.npz")
f5 = standin.build_model(fx5["model"]).fowtList[0]
tab5 = rq.pack_qtf(f5)
k5 = np.array([waves.wave_number(x, f5.depth) for x in w2])
n_k = 4
bet5 = np.array([0.0, 0.5, 1.0, 2.0])
for _ in range(2):
    ctx.qtf_kay([tab5] * n_k, bet5, w2, k5, f5.depth, f5.rho_water, f5.g)
ms_kay = ctx.last_kernel_ms()
t0 = time.perf_counter()
rq.kay_correction(tab5.kay_geom, w2, k5, 0.5, f5.depth, rho=f5.rho_water, g=f5.g)
t_host = time.perf_counter() - t0
print(json.dumps({"metric": "Kim & Yue table", "sets": n_k, "nw2": nw2, "items_per_set": int(len(rq.kay_items(tab5.kay_geom, 0.5))),
                  "device_ms_all_sets": ms_kay, "host_scipy_s_per_set": t_host,
                  "speedup_per_set": t_host / (ms_kay * 1e-3 / n_k)}))
