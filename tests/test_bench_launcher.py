"""CPU suite: `python bench.py --gpus N` starts its own N ranks (bench.py launch_ranks) -- against a stub rank
(tests/helpers/stub_rank.py) that checks the environment it is given and meets the other ranks over the host transport."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = "%s %s" % (sys.executable, os.path.join(ROOT, "tests", "helpers", "stub_rank.py"))


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RAFTX_COMM_TOKEN")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120)


def test_gpus_n_starts_n_ranks_without_a_launcher():
    r = _bench(["--gpus", "3", "--steps", "2"], RAFTX_BENCH_RANK_CMD=STUB)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # rank 0 alone owns stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["sum"] == 6.0 and out["argv"] == ["--gpus", "3", "--steps", "2"]


def test_a_failing_rank_stops_the_job_with_its_exit_code():
    t0 = time.time()
    r = _bench(["--gpus", "2"], RAFTX_BENCH_RANK_CMD=STUB, STUB_MODE="fail")
    assert r.returncode == 3 and "rank 1 exited with code 3" in r.stderr
    assert time.time() - t0 < 30                           # rank 0 (sleeping) was terminated, not waited for


def test_gpus_must_match_the_world_an_external_launcher_made():
    r = _bench(["--gpus", "8"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "--gpus 8 but WORLD_SIZE=2" in r.stderr
    r = _bench(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2


def test_more_gpus_than_the_host_has_fails_loudly():
    r = _bench(["--gpus", "2"])                            # this container has no GPU at all
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_through_bench_itself():
    """`python bench.py --gpus 2` with no launcher around it, both ranks on device 0 (RAFTX_BENCH_DEVICE=0: RCCL refuses two
    ranks on one device, the exchange steps take the host transport and the line says so): two ranks are started, the
    statistics of both are gathered inside the timed steps, rank 0 prints ONE line that says n_gpus = 2 and carries the
    per-rank / gather / single-rank keys."""
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--designs", "512", "--no-cpu-baseline"], RAFTX_BENCH_DEVICE="0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-500:]     # the JSON line is the LAST line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["designs_per_gpu"] == 512
    assert "host-tcp" in out["config"]["gather"]
    assert len(out["per_rank_ms"]["all"]) == 2 and out["gather_ms"]["rank0"] >= 0.0
    assert 0.2 < out["scaling_efficiency"] < 1.2 and out["single_rank_same_invocation"]["value"] > 0
    assert out["parity"]["niter_mismatches_vs_reference"] == 0
    # N > 1 weak line ALSO carries configs[2]'s literal shape: ONE sweep of --designs cut into N shards, same invocation
    st = out["strong_same_invocation"]
    assert st["scaling"] == "strong" and st["shard_designs"] == [256, 256] and st["total_designs"] == 512
    assert abs(st["value"] - 512 * out["config"]["nw"] / (st["ms_per_step"] * 1e-3)) < 1e-6 * st["value"]


@pytest.mark.gpu
def test_single_gpu_line_carries_the_numbers_a_reader_needs():
    """beside `value` the line has the step-level roofline fraction, SURVEY 8d's literal step (responses
    downloaded) as a top-level key, and the 1/8 shard of the sweep (what one rank of an 8-GPU strong-scaling run does)."""
    r = _bench(["--steps", "4", "--warmup", "1", "--designs", "2048", "--no-cpu-baseline", "--legs", "xi,shard"], RAFTX_BENCH_XI_STEPS="6")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rf = out["roofline"]
    assert 0 < rf["step_frac"] <= rf["frac"] < 1 and rf["kernel_ms_per_step"] <= out["ms_per_step"]
    assert abs(rf["step_frac"] - rf["algorithmic_flops_per_step"] / (out["ms_per_step"] * 1e-3) / 1e12 / rf["peak"]) < 1e-9
    assert out["value_xi_out"] == out["xi_out"]["streamed_dcf_per_s"] > 0
    sh = out["shard_1250"]
    assert sh["designs_per_step"] == 256 and sh["ms_per_step"] > 0
    assert abs(sh["projected_8_gpu_strong_speedup"] - out["ms_per_step"] / sh["ms_per_step"]) < 1e-9


@pytest.mark.gpu
def test_sharded_qtf_workload_through_bench_itself():
    """--workload c5 with two ranks on one GPU: rows of every QTF interleaved over the ranks, one SUM-reduce onto rank 0; the
    reduced matrices are Hermitian and finite (asserted inside), the line carries both ranks' kernel times."""
    r = _bench(["--gpus", "2", "--workload", "c5", "--steps", "2", "--sets", "2"], RAFTX_BENCH_DEVICE="0")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["hermitian"] is True
    assert len(out["per_rank_ms"]["qtf_kernels"]) == 2


@pytest.mark.gpu
def test_strong_scaling_mode_cuts_one_sweep_into_ragged_shards():
    """--scaling strong: ONE sweep of --designs cut into N contiguous shards (1001 over 3 ranks = 334 + 334 + 333),
    gathered with ragged counts inside the timed steps; the line says "strong", names the shards and carries every rank's
    descriptor-expansion time."""
    r = _bench(["--gpus", "3", "--scaling", "strong", "--designs", "1001", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
               RAFTX_BENCH_DEVICE="0")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert out["config"]["shard_designs"] == [334, 334, 333] and out["config"]["total_designs"] == 1001
    assert abs(out["value"] - 1001 * out["config"]["nw"] / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert len(out["host_descriptor_ms_per_rank"]["all"]) == 3
    assert out["parity"]["niter_mismatches_vs_reference"] == 0
