"""SURVEY.md section 5 (sanitizers): the CPU oracle compiled with -fsanitize=address,undefined into a small C driver
(tests/csrc/oracle_sanitizer_driver.c) and run through the C-ABI on reference-built C3 variants: no out-of-bounds
access, no leak of per-call scratch, no undefined behaviour, and the same responses as the ordinary build."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from raft_amd import snapshot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sanitized_driver(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("san") / "oracle_san")
    cmd = ["gcc", "-std=c99", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fopenmp", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "csrc", "oracle_sanitizer_driver.c"),
           os.path.join(ROOT, "oracle", "raftx_oracle.c"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "asan" in r.stderr.lower() or "ubsan" in r.stderr.lower() or "sanitize" in r.stderr.lower():
            pytest.skip("sanitizer runtimes not installed: " + r.stderr[-300:])
        raise AssertionError(r.stderr[-2000:])
    return exe


def test_oracle_under_address_and_undefined_behaviour_sanitizers(sanitized_driver, tmp_path, oracle_ctx):
    fx = snapshot.load_fixture("c3_variants.npz")
    nD = 3
    off = np.asarray(fx["strip_offsets"], dtype=np.int64)[:nD + 1]
    strips = np.ascontiguousarray(np.asarray(fx["strips"], dtype=np.float64)[:off[-1]])
    w, k = np.asarray(fx["w"], dtype=np.float64), np.asarray(fx["k"], dtype=np.float64)
    # two sea states, the second with two headings worth of amplitudes folded in as a second heading
    z0 = np.asarray(fx["zeta"], dtype=np.float64).reshape(1, -1)[0]
    zeta = np.ascontiguousarray(np.stack([np.stack([z0, 0.5 * z0]), np.stack([0.7 * z0, 0.2 * z0])]))       # [2 cases, 2 heads, nw]
    beta = np.array([[0.0, 0.6], [0.3, 2.0]])
    nIter = int(fx["nIter"])
    path = str(tmp_path / "problem.bin")
    with open(path, "wb") as f:
        np.array([nD, off[-1], len(w), 2, 2, nIter], dtype=np.int64).tofile(f)
        off.tofile(f)
        strips.tofile(f)
        for name in ("M0", "B0", "C0"):
            np.ascontiguousarray(np.asarray(fx[name], dtype=np.float64)[:nD]).tofile(f)
        w.tofile(f)
        k.tofile(f)
        np.array([float(fx["depth"]), float(fx["XiStart"])]).tofile(f)
        zeta.tofile(f)
        beta.tofile(f)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sanitized_driver, path], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.startswith("OK ")
    # the same call sequence through the ordinary build gives the same checksum
    o = oracle_ctx
    o.upload_designs_raw(off, strips, np.asarray(fx["M0"])[:nD], np.asarray(fx["B0"])[:nD], np.asarray(fx["C0"])[:nD], len(w))
    o.upload_cases(w, k, float(fx["depth"]), 1025.0, 9.81, zeta, beta)
    out = o.solve_dynamics(nIter, tol=0.01, XiStart=float(fx["XiStart"]), want_Xi=True)
    sd, _ = o.motion_stats(w[1] - w[0])
    cs = out["Xi"].real.sum() + out["Xi"].imag.sum() + out["niter"].sum() + sd[..., 0].sum()
    assert abs(float(r.stdout.split()[1]) - cs) <= 1e-9 * max(1.0, abs(cs))
