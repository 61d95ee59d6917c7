"""Host placement helper (raft_amd/locality.py): CPU-list parsing and the best-effort rules.  No GPU: the library is
replaced by a stand-in that reports a chosen NUMA node, and sysfs by a temporary directory."""
import os

import pytest

from raft_amd import backend, locality


class _Lib:
    def __init__(self, node, fail=False):
        self.node, self.fail = node, fail

    def device_locality(self, device_id=0):
        if self.fail:
            raise RuntimeError("no such device")
        return "0000:05:00.0", self.node


def _fake_sysfs(tmp_path, node, cpulist):
    d = tmp_path / ("node%d" % node)
    d.mkdir()
    (d / "cpulist").write_text(cpulist + "\n")
    return str(tmp_path)


def test_cpulist_parsing():
    assert locality._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert locality._parse_cpulist("") == set()
    assert locality.node_cpus(0, "/nonexistent") == set()


def test_unknown_topology_leaves_the_process_alone(tmp_path):
    before = os.sched_getaffinity(0)
    for lib in (_Lib(-1), _Lib(0, fail=True), _Lib(3)):              # node 3 has no cpulist in the fake sysfs
        rec = locality.bind_near_device(lib, 0, sysfs=str(tmp_path))
        assert not rec["bound"] and rec["why"]
        assert os.sched_getaffinity(0) == before


def test_binds_to_the_nodes_allowed_cpus_and_reports(tmp_path):
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip("needs two allowed CPUs")
    keep = sorted(before)[: len(before) // 2]
    sysfs = _fake_sysfs(tmp_path, 1, ",".join(str(c) for c in keep) + ",100000")   # a CPU we may not use is ignored
    try:
        rec = locality.bind_near_device(_Lib(1), 0, sysfs=sysfs)
        assert rec["bound"] and rec["cpus"] == len(keep) and rec["numa_node"] == 1
        assert os.sched_getaffinity(0) == set(keep)
        again = locality.bind_near_device(_Lib(1), 0, sysfs=sysfs)                 # second call: nothing left to do
        assert not again["bound"] and again["why"] == "already confined to the node"
    finally:
        os.sched_setaffinity(0, before)


def test_oracle_reports_no_locality(oracle_lib):
    assert oracle_lib.device_locality(0) == ("", -1)


@pytest.mark.gpu
def test_device_reports_a_pci_address():
    saved = os.sched_getaffinity(0)
    try:
        pci, node = backend.hip_library().device_locality(0)
        assert len(pci.split(":")) == 3 and node >= -1
        rec = locality.bind_near_device(backend.hip_library(), 0)
        assert rec["pci"] == pci
    finally:
        os.sched_setaffinity(0, saved)
